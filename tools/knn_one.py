"""One Hamming 2-NN shape, a few searches (for profiler runs): knn_one.py <nq> <nt> [mode] [waves]"""
import os
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from visual_odometry_amd import capi

nq, nt = int(sys.argv[1]), int(sys.argv[2])
ctx = capi.Context(0)
if len(sys.argv) > 3:
    ctx.set_option("knn_mode", int(sys.argv[3]))
if len(sys.argv) > 4:
    ctx.set_option("knn_waves", int(sys.argv[4]))
rng = np.random.default_rng(5)
q = torch.from_numpy(rng.integers(0, 256, (nq, 32), dtype=np.uint8)).cuda()
t = torch.from_numpy(rng.integers(0, 256, (nt, 32), dtype=np.uint8)).cuda()
idx = torch.zeros((nq, 2), dtype=torch.int32, device="cuda")
dist = torch.zeros((nq, 2), dtype=torch.int32, device="cuda")
for rep in range(2):
    ctx.hamming_knn2_dev(q.data_ptr(), nq, t.data_ptr(), nt, idx.data_ptr(), dist.data_ptr())
ctx.sync()
ctx.set_timing(True, ["knn2"])
for rep in range(5):
    ctx.hamming_knn2_dev(q.data_ptr(), nq, t.data_ptr(), nt, idx.data_ptr(), dist.data_ptr())
ctx.sync()
ms, n = ctx.kernel_time_ms(capi.T_GROUPS.index("knn2"))
print(f"{nq} x {nt}: {ms:.4f} ms per search")
