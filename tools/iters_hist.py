"""Development aid: histogram of the adaptive RANSAC iteration counts on the sparse benchmark scene."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from visual_odometry_amd import capi, synth

P = 128
frames = []
for p in range(P):
    frames += list(synth.frame_pair(0x5EED0000 + p, sparse=True, moving_fraction=0.6, n_layers=24) if "--dense" not in sys.argv else synth.frame_pair(0x5EED0000 + p))
imgs = torch.from_numpy(np.stack(frames)).cuda()
ctx = capi.Context(0)
res = torch.zeros(P * capi.PAIR_DTYPE.itemsize, dtype=torch.uint8, device="cuda")
K = synth.default_K(640, 480)
ctx.pipeline_pairs_dev(imgs.data_ptr(), P, 480, 640, 640, 480 * 640, 2000, 0.75, K, 1.0, 0.999, 0, res.data_ptr())
ctx.sync()
r = np.frombuffer(res.cpu().numpy().tobytes(), capi.PAIR_DTYPE)
it = r["ransac_iters"]
print("iters: mean %.0f" % it.mean(), "hist", np.histogram(it, bins=[0, 33, 129, 257, 513, 641, 769, 897, 1000, 1001])[0].tolist())
print("sorted", np.sort(it).tolist())
