#!/bin/bash
# rocprofv3 kernel statistics of one benchmark workload: tools/profile_workload.sh <workload> <tag> [extra bench flags]
export TMPDIR=/tmp
R=$PWD
W=$1; TAG=$2; shift 2
rm -rf $R/gpurun_out/prof_$TAG
cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG -- python3 $R/bench.py --workload $W --no-cpu-baseline --no-extras --no-hamming --no-sequence --no-aux-scene --no-dropin --no-scene-counters --no-configs "$@" > $R/gpurun_out/prof_$TAG.json 2> $R/gpurun_out/prof_$TAG.err
