#!/bin/bash
# Texture-addresser / L1 counters of the pipeline kernels (one rocprofv3 pass).
set -e
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/pmc_mem
mkdir -p $OUT
cd /tmp
C="TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_GATE_EN1_sum TD_TD_BUSY_sum GRBM_GUI_ACTIVE"
timeout -s KILL 300 rocprofv3 --pmc $C --output-format csv -d $OUT/a -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --streams 1 --no-hamming --no-cpu-baseline --no-extras --no-sequence --no-aux-scene --no-dropin --no-scene-counters --no-configs
