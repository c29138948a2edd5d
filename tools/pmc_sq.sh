#!/bin/bash
# Instruction-mix and stall counters of the pipeline kernels, two rocprofv3 passes of eight
# SQ counters each.  Run on the GPU box from the repository root.
set -e
export TMPDIR=/tmp
# usage: tools/pmc_sq.sh [output directory under gpurun_out/ (default pmc_sq)] [extra bench.py flags]
OUT=$PWD/gpurun_out/${1:-pmc_sq}
shift || true
mkdir -p $OUT
cd /tmp
A="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_ANY"
B="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
timeout -s KILL 300 rocprofv3 --pmc $A --output-format csv -d $OUT/a -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --streams 1 --no-hamming --no-cpu-baseline --no-extras --no-sequence --no-aux-scene --no-dropin --no-scene-counters --no-configs "$@"
timeout -s KILL 300 rocprofv3 --pmc $B --output-format csv -d $OUT/b -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --streams 1 --no-hamming --no-cpu-baseline --no-extras --no-sequence --no-aux-scene --no-dropin --no-scene-counters --no-configs "$@"
