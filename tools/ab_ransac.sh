#!/bin/bash
# A/B timing of library builds on the RANSAC-heavy workloads: tools/ab_ransac.sh ab_libs/A.so ab_libs/B.so ...
set -e
S="--no-hamming --no-cpu-baseline --no-extras --no-sequence --no-aux-scene --no-dropin"
for round in 1 2; do
  for so in "$@"; do
    cp "$so" visual_odometry_amd/libvo_hip.so
    for w in "--scene sparse --distinct 64 --steps 8" "--workload c5_1080p_10000kp_4096hyp --steps 5" "--workload c1_640x480_1000kp_seq100 --steps 20"; do
      timeout -k 10 120 python3 bench.py $S $w > gpurun_out/ab_tmp.json 2> gpurun_out/ab_tmp.err
      python3 - "$so" "$w" <<'PY'
import json, sys
d = json.load(open("gpurun_out/ab_tmp.json"))
k = d["pipeline"]["kernel_group_ms_per_step_single_stream"]
print(sys.argv[1], sys.argv[2].split()[1], round(d["value"]), "ms/step %.3f" % d["ms_per_step"],
      " ".join(f"{g}={k[g]:.3f}" for g in ("ransac", "ransac_solve", "ransac_score", "ransac_other")), flush=True)
PY
    done
  done
done
