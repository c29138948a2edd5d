#!/bin/bash
# rocprofv3 kernel statistics of the default benchmark command (two streams).
export TMPDIR=/tmp
R=$PWD
rm -rf $R/gpurun_out/prof_default
cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_default -- python3 $R/bench.py --no-cpu-baseline --no-extras --no-aux-scene --no-dropin --no-scene-counters --no-configs > $R/gpurun_out/prof_default.log 2>&1
