#!/usr/bin/env python3
"""Per-kernel averages of the counters in rocprofv3 counter_collection.csv files.

    python tools/pmc_summary.py gpurun_out/pmc_sq/a gpurun_out/pmc_sq/b
"""
import collections
import csv
import glob
import os
import sys

table = collections.defaultdict(dict)
for d in sys.argv[1:]:
    files = sorted(glob.glob(f"{d}/**/*counter_collection.csv", recursive=True), key=os.path.getmtime)
    if not files:
        continue
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(files[-1])):
        agg[(r["Kernel_Name"].split("(")[0][-28:], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k, c), v in agg.items():
        table[k][c] = sum(v) / len(v)
cols = sorted({c for v in table.values() for c in v})
print("kernel".ljust(30) + " ".join(c.replace("SQ_", "")[:14].rjust(15) for c in cols))
for k, v in table.items():
    if "vo::" in k or "kernel" in k:
        print(k.ljust(30) + " ".join(f"{v.get(c, float('nan')):15.4g}" for c in cols))
