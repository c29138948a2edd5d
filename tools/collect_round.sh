#!/bin/bash
# Everything the profiles/ directory holds for one round, taken in one visit to the GPU box:
#   tools/collect_round.sh            (then: python tools/collect_profiles.py r02)
# Each GPU step has its own time limit and the chain stops at the first failure.
set -e
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out
mkdir -p $O
SHORT="--no-cpu-baseline --no-hamming --no-sequence --no-aux-scene --no-dropin --no-configs"

# (the counter calibration program is a built file that is not kept in history)
[ -x $R/tools/calib_fetch ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 $R/tools/calib_fetch.hip -o $R/tools/calib_fetch
echo "[collect] default bench line"; date +%T
timeout -k 10 400 python3 $R/bench.py > $O/bench_final.json 2> $O/bench_final.err
echo "[collect] rocprofv3 kernel statistics: default command"; date +%T
bash $R/tools/profile_default.sh
for spec in "c3_kitti_1241x376_4000kp c3" "c5_1080p_10000kp_4096hyp c5" "c1_640x480_1000kp_seq100 c1"; do
  set -- $spec
  echo "[collect] rocprofv3 kernel statistics: $1"; date +%T
  bash $R/tools/profile_workload.sh $1 $2 --steps 10 --warmup 2
done
echo "[collect] rocprofv3 kernel statistics: sparse scene"; date +%T
bash $R/tools/profile_workload.sh c2_640x480_2000kp sparse --scene sparse --steps 10 --warmup 2
echo "[collect] serial kernel split at C2 and C5"; date +%T
bash $R/tools/serial_kernel_times.sh > $O/serial_c2.txt
bash $R/tools/serial_kernel_times.sh --workload c5_1080p_10000kp_4096hyp > $O/serial_c5.txt
bash $R/tools/serial_kernel_times.sh --scene sparse > $O/serial_sparse.txt
echo "[collect] HBM traffic counters"; date +%T
bash $R/tools/pmc_traffic.sh > $O/pmc_traffic.log 2>&1
echo "[collect] SQ counters"; date +%T
bash $R/tools/pmc_sq.sh > $O/pmc_sq.log 2>&1
echo "[collect] SQ counters: sparse scene (the RANSAC kernels under load)"; date +%T
bash $R/tools/pmc_sq.sh pmc_sq_sparse --scene sparse > $O/pmc_sq_sparse.log 2>&1
echo "[collect] two ranks sharing the GPU (gloo): the multi-rank code path"; date +%T
cd $R && VO_BENCH_SHARE_GPUS=1 timeout -k 10 400 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 10 --warmup 2 --no-configs > $O/rehearsal_2ranks.json 2> $O/rehearsal_2ranks.err
echo "[collect] done"; date +%T
