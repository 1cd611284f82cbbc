#!/usr/bin/env python3
"""Turn a rocprofv3 --kernel-trace --stats output directory into a small tracked summary under profiles/.
usage: tools/summarize_prof.py <rocprof dir> <profiles/name.md> [steps_in_run] [bench json line file]"""
import csv
import glob
import sys

src, dst = sys.argv[1], sys.argv[2]
steps = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
f = glob.glob(src + "/**/*_kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
with open(dst, "w") as o:
    o.write(f"# rocprofv3 --kernel-trace --stats summary\n\nsource: `{f.split('gpurun_out/')[-1]}`; passes in run: {steps:g} "
            f"(warm-up + timed + instrumented); total kernel time {tot / 1e6:.2f} ms = {tot / 1e6 / steps:.2f} ms per pass\n\n")
    if len(sys.argv) > 4:
        lines = [l for l in open(sys.argv[4]).read().splitlines() if l.startswith("{")]
        o.write("bench line of the same run:\n\n```\n" + (lines[-1] if lines else "(no JSON line found)") + "\n```\n\n")
    o.write("| kernel | calls | avg µs | min µs | max µs | ms per pass | % |\n|---|---|---|---|---|---|---|\n")
    for r in rows:
        if float(r["TotalDurationNs"]) / tot < 5e-4:
            continue
        o.write(f"| `{r['Name'][:110]}` | {r['Calls']} | {float(r['AverageNs']) / 1e3:.1f} | {float(r['MinNs']) / 1e3:.1f} | "
                f"{float(r['MaxNs']) / 1e3:.1f} | {float(r['TotalDurationNs']) / 1e6 / steps:.3f} | {100 * float(r['TotalDurationNs']) / tot:.1f} |\n")
print("wrote", dst)
