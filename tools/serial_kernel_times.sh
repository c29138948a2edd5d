#!/bin/bash
# Exclusive per-kernel times: rocprofv3 kernel trace of a short bench run, reduced to the bench's
# own serial pass (three pipeline calls with every group bracketed, one stream, a sync between calls).
# Extra arguments go to bench.py (e.g. --workload c5_1080p_10000kp_4096hyp).
export TMPDIR=/tmp
R=$PWD
rm -rf $R/gpurun_out/prof_serial
cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_serial -- python3 $R/bench.py --no-cpu-baseline --no-extras --no-hamming --no-sequence --no-aux-scene --no-dropin --no-scene-counters --no-configs --warmup 1 --steps 2 "$@" > $R/gpurun_out/prof_serial.log 2>&1 || exit 1
python3 - $(ls $R/gpurun_out/prof_serial/*/*kernel_trace.csv | head -1) <<'PY'
import csv, sys, collections
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
per = collections.defaultdict(list)
for r in rows:
    per[r["Kernel_Name"].split("(")[0].split("::")[-1]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
tot = 0.0
out = []
for k, v in per.items():
    if len(v) % 9:
        continue
    n = len(v) // 9                      # launches per pipeline call: 1 warm-up, then the serial pass
    ser = v[n:4 * n]                     # (calls 1..3), 3 calls of the RANSAC split, 2 timed steps
    out.append((sum(ser) / 3e3, n, k))
    tot += sum(ser) / 3e3
for us, n, k in sorted(out, reverse=True):
    print(f"{k[:36]:36s} {n:3d} launches/call {us:9.1f} us/call {100 * us / tot:5.1f} %")
print("total", round(tot, 1), "us per call")
PY
