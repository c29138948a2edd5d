#!/bin/bash
# SQ counters (two passes of eight) of every kernel of one bench.py run: tools/pmc_bench.sh <bench flags...>
export TMPDIR=/tmp
R=$PWD
OUT=$R/gpurun_out/pmc_bench
rm -rf $OUT; mkdir -p $OUT
cd /tmp
A="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_ANY"
B="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR SQ_INSTS_FLAT"
S="--no-hamming --no-cpu-baseline --no-extras --no-sequence --no-aux-scene --no-dropin --no-scene-counters --no-configs --steps 2 --warmup 1"
timeout -s KILL 200 rocprofv3 --pmc $A --output-format csv -d $OUT/a -- python3 $R/bench.py $S "$@" > $OUT/a.log 2>&1 || exit 1
timeout -s KILL 200 rocprofv3 --pmc $B --output-format csv -d $OUT/b -- python3 $R/bench.py $S "$@" > $OUT/b.log 2>&1 || exit 1
python3 $R/tools/pmc_summary.py $OUT/a $OUT/b
