"""A/B of the Hamming 2-NN kernels (VO_OPT_KNN_MODE: 1 vector ALUs, 2 matrix cores): equality of the
tables on ragged, duplicate-laden and split cases, and time per search at several sizes."""
import os
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from visual_odometry_amd import capi

ctx = capi.Context(0)
rng = np.random.default_rng(11)


def run(mode, q, t, waves=0, split=0):
    ctx.set_option("knn_mode", mode)
    ctx.set_option("knn_waves", waves)
    ctx.set_option("knn_train_split", split)
    return ctx.hamming_knn2(q, t)


bad = 0
cases = [(1, 1), (1, 2), (5, 3), (64, 64), (65, 33), (100, 31), (100, 32), (257, 1000), (2000, 2000), (31, 4097),
         (700, 30720), (300, 30721), (130, 70000)]
for nq, nt in cases:
    t = rng.integers(0, 256, (nt, 32), dtype=np.uint8)
    q = rng.integers(0, 256, (nq, 32), dtype=np.uint8)
    # near duplicates and exact copies: ties in both places
    for k in range(0, nq, 3):
        q[k] = t[rng.integers(0, nt)]
    if nt > 8:
        t[nt // 2] = t[1]
        t[nt - 1] = t[1]
        t[nt // 3] = t[0]
        q[0] = t[1]
    for waves, split in ((0, 0), (1, 0), (4, 0), (8, 0), (8, 3), (4, 5)):
        ref = run(1, q, t, waves, split)
        got = run(2, q, t, waves, split)
        same = all(np.array_equal(a, b) for a, b in zip(ref, got))
        if not same:
            bad += 1
            i = np.flatnonzero((ref[0] != got[0]).any(1) | (ref[1] != got[1]).any(1))
            print("MISMATCH", nq, nt, waves, split, "rows", i[:5], ref[0][i[:3]], got[0][i[:3]], ref[1][i[:3]], got[1][i[:3]])
print("cases wrong:", bad, flush=True)

ctx.set_option("knn_waves", 0)
ctx.set_option("knn_train_split", 0)
for nq, nt in ((65536, 65536), (8192, 65536), (2000, 2000), (10000, 10000)):
    q = torch.from_numpy(rng.integers(0, 256, (nq, 32), dtype=np.uint8)).cuda()
    t = torch.from_numpy(rng.integers(0, 256, (nt, 32), dtype=np.uint8)).cuda()
    idx = torch.zeros((nq, 2), dtype=torch.int32, device="cuda")
    dist = torch.zeros((nq, 2), dtype=torch.int32, device="cuda")
    for mode in (1, 2):
        ctx.set_option("knn_mode", mode)
        for rep in range(2):
            ctx.hamming_knn2_dev(q.data_ptr(), nq, t.data_ptr(), nt, idx.data_ptr(), dist.data_ptr())
        ctx.sync()
        ctx.set_timing(True, ["knn2"])
        for rep in range(5):
            ctx.hamming_knn2_dev(q.data_ptr(), nq, t.data_ptr(), nt, idx.data_ptr(), dist.data_ptr())
        ctx.sync()
        ms, n = ctx.kernel_time_ms(capi.T_GROUPS.index("knn2"))
        ctx.set_timing(False)
        print(f"{nq} x {nt} mode {mode}: {ms:.4f} ms per search ({n} launches)", flush=True)
