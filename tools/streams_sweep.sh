#!/bin/bash
# headline step time with 1, 2 and 3 contexts (streams) rotating over the steps
for n in 1 2 3; do
  timeout -k 10 200 python bench.py --streams $n --no-cpu-baseline --no-extras --no-configs --no-hamming --no-sequence --no-aux-scene --no-dropin "$@" > gpurun_out/streams.json 2> gpurun_out/streams.err || exit 1
  python - <<PY
import json
d = json.load(open("gpurun_out/streams.json"))
print("streams $n:", round(d["value"]), "pairs/s", round(d["ms_per_step"], 3), "ms/step")
PY
done
