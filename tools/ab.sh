#!/bin/bash
# A/B timing of library builds on one GPU box: tools/ab.sh build_ab/A.so build_ab/B.so ...
# Runs the default pipeline bench twice per build, alternating, and prints pairs/s and the
# per-group kernel times.  (The box's tree is a scratch copy, so replacing the library is safe.)
set -e
for round in 1 2; do
  for so in "$@"; do
    cp "$so" visual_odometry_amd/libvo_hip.so
    timeout -k 10 120 python3 bench.py --no-hamming --no-cpu-baseline --no-extras --no-sequence --steps 40 > gpurun_out/ab_tmp.json 2> gpurun_out/ab_tmp.err
    python3 - "$so" <<'PY'
import json, sys
d = json.load(open("gpurun_out/ab_tmp.json"))
k = d["pipeline"]["kernel_group_ms_per_step_single_stream"]
print(sys.argv[1], round(d["value"]), " ".join(f"{g}={v:.3f}" for g, v in k.items()), flush=True)
PY
  done
done
