#!/usr/bin/env python3
"""Per-kernel HBM traffic from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE), corrected as
/opt/skills/guides/MI355X_MICROARCH.md §HBM prescribes: counters are in KiB; on gfx950 FETCH_SIZE reports exactly half
of the bytes of wide coalesced streaming reads (16 B/lane) -> doubled; WRITE_SIZE is taken as is (exact for 16-B/lane
stores; our conv epilogue stores 8 B/lane, which the guide calls uncalibrated — flagged in the output).
usage: tools/summarize_pmc.py <fetch dir> <write dir> <out.json> [<out.md>]"""
import csv, glob, json, re, sys
fd, wd, outj = sys.argv[1:4]
def load(d, name):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    acc = {}
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != name:
            continue
        m = re.search(r"(k_\w+)(<[^>]*>)?", r["Kernel_Name"])
        key = m.group(0).replace(" ", "") if m else r["Kernel_Name"][:60]
        a = acc.setdefault(key, [0.0, 0])
        a[0] += float(r["Counter_Value"]); a[1] += 1
    return acc
F, W = load(fd, "FETCH_SIZE"), load(wd, "WRITE_SIZE")
out = {}
for k in sorted(set(F) | set(W)):
    f, nf = F.get(k, [0, 1]); w, nw = W.get(k, [0, 1])
    out[k] = {"launches": max(nf, nw), "fetch_bytes_per_launch": 2.0 * 1024 * f / max(nf, 1),
              "write_bytes_per_launch": 1024.0 * w / max(nw, 1)}
    out[k]["hbm_bytes_per_launch"] = out[k]["fetch_bytes_per_launch"] + out[k]["write_bytes_per_launch"]
import subprocess
commit = subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip() or None
json.dump({"commit": commit, "note": "FETCH_SIZE x2 (gfx950 correction), WRITE_SIZE as is; KiB -> bytes; averages over all launches of the kernel "
                   "in `python bench.py --steps 2 --warmup 1 --no-graph` (batch 32, 506x506, bf16 / mixed); `commit` = HEAD when the summary was written", "kernels": out},
          open(outj, "w"), indent=1)
if len(sys.argv) > 4:
    with open(sys.argv[4], "w") as o:
        o.write("# HBM traffic per launch (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes)\n\n"
                "FETCH_SIZE doubled (gfx950 correction), KiB -> bytes.\n\n| kernel | launches | fetch MB | write MB | total MB |\n|---|---|---|---|---|\n")
        for k, v in sorted(out.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"] * kv[1]["launches"]):
            o.write(f"| `{k}` | {v['launches']} | {v['fetch_bytes_per_launch'] / 1e6:.1f} | {v['write_bytes_per_launch'] / 1e6:.1f} | {v['hbm_bytes_per_launch'] / 1e6:.1f} |\n")
print("wrote", outj)
