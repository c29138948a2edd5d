#!/bin/bash
# HBM traffic of the pipeline kernels from the L2 memory-side counters: one rocprofv3 pass per
# counter (FETCH_SIZE and WRITE_SIZE do not fit one pass), plus the same two passes over a
# calibration program with known byte counts.  Run on the GPU box from the repository root.
set -e
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/pmc_traffic
mkdir -p $OUT
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 120 rocprofv3 --pmc $c --output-format csv -d $OUT/calib_$c -- $GRAFT_REPO_ROOT/tools/calib_fetch
  timeout -s KILL 300 rocprofv3 --pmc $c --output-format csv -d $OUT/bench_$c -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --streams 1 --no-hamming --no-cpu-baseline --no-extras --no-sequence --no-aux-scene --no-dropin --no-scene-counters --no-configs
done
