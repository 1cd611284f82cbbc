#!/usr/bin/env python3
"""Timeline of ONE training step from a rocprofv3 --kernel-trace CSV: per queue busy time, the gaps on the main queue and
the kernels of the step in start order.  usage: tools/step_timeline.py <run_kernel_trace.csv> [step index from the end] [-v]"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
starts = [i for i, r in enumerate(rows) if "k_pack_nchw" in r["Kernel_Name"]]
k = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2].lstrip("-").isdigit() else 3
a, b = starts[-k - 1], starts[-k]
step = rows[a:b]
t0 = step[0]["s"]
print(f"step of {len(step)} kernels, {(step[-1]['e'] - t0) / 1e3:.1f} us from first start to last end")
def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    return re.sub(r"\(.*", "", n)[:48]
byq = {}
for r in step:
    byq.setdefault(r["Queue_Id"], []).append(r)
for q, rs in byq.items():
    busy = sum(r["e"] - r["s"] for r in rs)
    print(f"queue {q}: {len(rs)} kernels, busy {busy / 1e3:.1f} us, span {(rs[-1]['e'] - rs[0]['s']) / 1e3:.1f} us")
main = max(byq.values(), key=len)
gaps = sorted(((main[i + 1]["s"] - main[i]["e"]) / 1e3, short(main[i]["Kernel_Name"]), short(main[i + 1]["Kernel_Name"])) for i in range(len(main) - 1))
print("main-queue idle between kernels: total %.1f us; largest:" % sum(g[0] for g in gaps if g[0] > 0))
for g in gaps[-8:]:
    print("   %.1f us after %s before %s" % g)
agg = {}
for r in step:
    d = agg.setdefault((r["Queue_Id"], short(r["Kernel_Name"])), [0, 0.0])
    d[0] += 1; d[1] += (r["e"] - r["s"]) / 1e3
print("per kernel (queue, name): calls, total us")
for (q, n), (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:28]:
    print(f"  q{q} {n:48s} {c:4d} {t:9.1f}")
if "-v" in sys.argv:
    for r in step:
        print(f"{(r['s'] - t0) / 1e3:9.1f} {(r['e'] - r['s']) / 1e3:8.1f} q{r['Queue_Id']} {short(r['Kernel_Name'])} grid {r['Grid_Size_X']}x{r['Grid_Size_Y']}x{r['Grid_Size_Z']}")
