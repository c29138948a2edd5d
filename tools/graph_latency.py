"""Repeated small pipeline calls with and without VO_OPT_PIPELINE_GRAPH: time per call (a fence after
every call, and back to back) and equality of the result records."""
import os
import sys
import time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from visual_odometry_amd import capi, synth

ctx = capi.Context(0)
K = synth.TUM_K
for pairs in (1, 2, 8, 32):
    frames = []
    for k in range(pairs):
        frames.extend(synth.frame_pair(0x5EED0000 + k))
    imgs = torch.from_numpy(np.stack(frames)).cuda()
    res = torch.zeros(pairs * capi.PAIR_DTYPE.itemsize, dtype=torch.uint8, device="cuda")

    def call():
        ctx.pipeline_pairs_dev(imgs.data_ptr(), pairs, 480, 640, 640, 480 * 640, 2000, 0.75, K, 1.0, 0.999, 0, res.data_ptr())

    out = {}
    for graph in (0, 1):
        ctx.set_option("pipeline_graph", graph)
        for _ in range(5):
            call()
        ctx.sync()
        out[graph] = res.cpu().numpy().tobytes()
        n = 300
        t0 = time.perf_counter()
        for _ in range(n):
            call()
            ctx.sync()
        fenced = (time.perf_counter() - t0) / n
        t0 = time.perf_counter()
        for _ in range(n):
            call()
        ctx.sync()
        burst = (time.perf_counter() - t0) / n
        print(f"{pairs:3d} pairs, graph {graph}: {1e3 * fenced:.4f} ms per fenced call, {1e3 * burst:.4f} ms per call back to back", flush=True)
    print("    records equal:", out[0] == out[1], flush=True)
ctx.set_option("pipeline_graph", 0)
