"""A few single-pair pipeline calls with a fence after each (run under rocprofv3 --kernel-trace;
tools/one_pair_trace.sh reduces the trace to the last call's kernel chain)."""
import os
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from visual_odometry_amd import capi, synth

pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 1
sparse = len(sys.argv) > 2 and sys.argv[2] == "sparse"
frames = []
for k in range(pairs):
    frames.extend(synth.frame_pair(0x5EED0000 + k, sparse=True, moving_fraction=0.6, n_layers=24) if sparse else synth.frame_pair(0x5EED0000 + k))
imgs = torch.from_numpy(np.stack(frames)).cuda()
ctx = capi.Context(0)
res = torch.zeros(pairs * capi.PAIR_DTYPE.itemsize, dtype=torch.uint8, device="cuda")
for rep in range(6):
    ctx.pipeline_pairs_dev(imgs.data_ptr(), pairs, 480, 640, 640, 480 * 640, 2000, 0.75, synth.TUM_K, 1.0, 0.999, 0, res.data_ptr())
    ctx.sync()
out = np.frombuffer(res.cpu().numpy().tobytes(), capi.PAIR_DTYPE)
print("iters", out["ransac_iters"], "inliers", out["inliers"])
