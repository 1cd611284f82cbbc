#!/usr/bin/env python3
"""Compact per-kernel resource table (VGPRs, spills, LDS, occupancy) of one .hip source, from hipcc's
-Rpass-analysis=kernel-resource-usage remarks.  usage: tools/kernel_resources.py csrc/<file>.hip [name filter]"""
import re, subprocess, sys
src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
cmd = ["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++17", "-ffp-contract=off", *sys.argv[3:],
       "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/tmp/_kr.o"] + (["-fno-slp-vectorize"] if "conv_rr" in src else [])
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = {}
rows = []
for line in out.splitlines():
    m = re.search(r"remark: [^:]*:\d+:\d+: +(.*?) \[-Rpass", line) or re.search(r"remark: +(.*?) \[-Rpass", line)
    if not m:
        m = re.search(r":\d+:\d+: remark: (.*?) \[-Rpass", line)
    if not m:
        continue
    t = m.group(1).strip()
    if t.startswith("Function Name:"):
        cur = {"name": t.split(":", 1)[1].strip()}
        rows.append(cur)
    elif ":" in t and cur is not None:
        k, v = t.split(":", 1)
        cur[k.strip()] = v.strip()
for r in rows:
    name = subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.strip()
    name = re.sub(r"\(ConvGeom.*", "", name).replace("(anonymous namespace)::", "").replace("void ", "")
    if flt and flt not in name:
        continue
    print(f"{name[:70]:70s} VGPR {r.get('VGPRs','?'):>4} AGPR {r.get('AGPRs','?'):>3} spill {r.get('VGPR Spill', r.get('VGPRs Spill','?')):>3} "
          f"SGPR {r.get('SGPRs', r.get('TotalSGPRs','?')):>4} scratch {r.get('ScratchSize [bytes/lane]','?'):>4} occ {r.get('Occupancy [waves/SIMD]','?')} LDS {r.get('LDS Size [bytes/block]','?')}")
