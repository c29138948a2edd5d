"""One Hamming 2-NN shape under several launch variants (also for profiler runs):
    knn_one.py <nq> <nt> [mode] [waves,waves,...] [split,split,...]"""
import os
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from visual_odometry_amd import capi

nq, nt = int(sys.argv[1]), int(sys.argv[2])
mode = int(sys.argv[3]) if len(sys.argv) > 3 else -1
waves = [int(v) for v in sys.argv[4].split(",")] if len(sys.argv) > 4 else [0]
splits = [int(v) for v in sys.argv[5].split(",")] if len(sys.argv) > 5 else [0]
ctx = capi.Context(0)
ctx.set_option("knn_mode", mode)
rng = np.random.default_rng(5)
q = torch.from_numpy(rng.integers(0, 256, (nq, 32), dtype=np.uint8)).cuda()
t = torch.from_numpy(rng.integers(0, 256, (nt, 32), dtype=np.uint8)).cuda()
idx = torch.zeros((nq, 2), dtype=torch.int32, device="cuda")
dist = torch.zeros((nq, 2), dtype=torch.int32, device="cuda")
for w in waves:
    for sp in splits:
        ctx.set_option("knn_waves", w)
        ctx.set_option("knn_train_split", sp)
        for rep in range(2):
            ctx.hamming_knn2_dev(q.data_ptr(), nq, t.data_ptr(), nt, idx.data_ptr(), dist.data_ptr())
        ctx.sync()
        ctx.set_timing(True, ["knn2"])
        for rep in range(5):
            ctx.hamming_knn2_dev(q.data_ptr(), nq, t.data_ptr(), nt, idx.data_ptr(), dist.data_ptr())
        ctx.sync()
        ms, n = ctx.kernel_time_ms(capi.T_GROUPS.index("knn2"))
        ctx.set_timing(False)
        print(f"{nq} x {nt} mode {mode} waves {w} split {sp}: {ms * n / 5:.4f} ms per search", flush=True)
