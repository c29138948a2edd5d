#!/bin/bash
# RANSAC work loop on the GPU box: the RANSAC parity tests, then the sparse-scene and C5 benches.
# usage: tools/ransac_round.sh <tag>
tag=$1
timeout -k 10 600 python -u -m pytest tests -m gpu -q --timeout 300 --timeout-method thread -k "two_view or five_point or pure_translation or motion_scenes or essential or sample_stream or estimate_motion or full_hd or c1_sequence or pipeline_pairs_matches or large_rounds or variants_agree" > gpurun_out/${tag}_tests.log 2>&1
rc=$?
tail -3 gpurun_out/${tag}_tests.log
[ $rc -ne 0 ] && grep -E "^E |Error" gpurun_out/${tag}_tests.log | head -20
[ $rc -gt 1 ] && exit $rc
timeout -k 10 200 python bench.py --scene sparse --no-cpu-baseline --no-extras > gpurun_out/${tag}_bench_sparse.json 2> gpurun_out/${tag}_bench.err || exit 1
timeout -k 10 200 python bench.py --workload c5_1080p_10000kp_4096hyp --no-cpu-baseline --no-extras > gpurun_out/${tag}_bench_c5.json 2>> gpurun_out/${tag}_bench.err || exit 1
python - <<PY
import json
for f in ("sparse", "c5"):
    d = json.load(open("gpurun_out/${tag}_bench_%s.json" % f))
    g = d["pipeline"]["kernel_group_ms_per_step_single_stream"]
    print(f, round(d["value"]), "pairs/s", round(d["ms_per_step"], 3), "ms/step; ransac", {k: round(v, 3) for k, v in g.items() if "ransac" in k})
PY
