#!/usr/bin/env python3
"""Durations and gaps of the kernels of the last pipeline step in a rocprofv3 kernel trace."""
import csv, glob, os, sys
d = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/trace1"
f = sorted(glob.glob(f"{d}/**/*kernel_trace.csv", recursive=True), key=os.path.getmtime)[-1]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "resize" in r["Kernel_Name"]]
prev = None
for r in rows[idx[-7]:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(r["Kernel_Name"][:34].ljust(34), f"{(e - s) / 1e3:8.1f} us  gap {((s - prev) if prev else 0) / 1e3:6.1f}")
    prev = e
