#!/bin/bash
# Kernel chain of one small pipeline call: start offset, duration and gap of every kernel of the last of six calls.
# usage: tools/one_pair_trace.sh [pairs] [sparse]
export TMPDIR=/tmp
R=$PWD
rm -rf $R/gpurun_out/prof_one
cd /tmp && timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_one -- python3 $R/tools/one_pair_trace.py "$@" > $R/gpurun_out/prof_one.log 2>&1 || exit 1
python3 - $(ls $R/gpurun_out/prof_one/*/*kernel_trace.csv | head -1) <<'PY'
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
n = len(rows) // 6
last = rows[-n:]
t0 = int(last[0]["Start_Timestamp"])
prev_end = t0
busy = 0
for r in last:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].split("(")[0].split("::")[-1][:34]
    print(f"{(s - t0) / 1e3:8.1f} us  {name:34s} {(e - s) / 1e3:7.1f} us  gap {(s - prev_end) / 1e3:6.1f}")
    prev_end = max(prev_end, e)
    busy += e - s
print("kernels", n, "chain", (prev_end - t0) / 1e3, "us; sum of kernel durations", busy / 1e3, "us")
PY
