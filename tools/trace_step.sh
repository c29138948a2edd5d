#!/bin/bash
# Kernel trace of a short single-stream run; tools/trace_step.py prints the last step's kernels.
export TMPDIR=/tmp
R=$PWD
rm -rf $R/gpurun_out/trace1
cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/trace1 -- python3 $R/bench.py --steps 4 --warmup 1 --streams 1 --no-hamming --no-cpu-baseline --no-extras > $R/gpurun_out/trace1.log 2>&1
