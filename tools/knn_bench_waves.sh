#!/bin/bash
# knn2 group time of a bench workload under several wavefront counts per workgroup: tools/knn_bench_waves.sh <workload> <nw>...
W=$1; shift
for nw in "$@"; do
  VO_KNN_NW=$nw timeout -k 10 200 python bench.py --workload $W --no-cpu-baseline --no-extras --no-configs --no-hamming --no-sequence --no-aux-scene --no-dropin --steps 10 --warmup 2 > gpurun_out/knn_nw.json 2> gpurun_out/knn_nw.err || exit 1
  python - <<PY
import json
d = json.load(open("gpurun_out/knn_nw.json"))
print("$W nw=$nw:", round(d["value"]), "pairs/s; knn2", round(d["pipeline"]["kernel_group_ms_per_step_single_stream"]["knn2"], 4), "ms")
PY
done
