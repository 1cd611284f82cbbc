#!/usr/bin/env python3
"""Average PMC counter values per kernel from the rocprofv3 csv output of tools/pmc_conv.sh.
usage: tools/pmc_table.py gpurun_out/pmc_fwd_0 [gpurun_out/pmc_fwd_1 ...]"""
import csv, glob, sys
from collections import defaultdict
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        acc = defaultdict(lambda: defaultdict(list))
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, cs in acc.items():
            if "mfma" not in k and "conv_rr" not in k and "wgrad" not in k and "gn_" not in k and "bicubic" not in k:
                continue
            print(k)
            for c, v in cs.items():
                print(f"   {c:34s} {sum(v) / len(v):16.1f}   (n={len(v)})")
