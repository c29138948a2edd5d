"""Fenced single-pair pipeline calls under option settings given as name=value,... groups:
    one_pair_variants.py "" "solve_variant=1" "second_chart=0" ...   (time per call; records compared with the first group)"""
import os
import sys
import time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from visual_odometry_amd import capi, synth

ctx = capi.Context(0)
frames = list(synth.frame_pair(0x5EED0000))
imgs = torch.from_numpy(np.stack(frames)).cuda()
res = torch.zeros(capi.PAIR_DTYPE.itemsize, dtype=torch.uint8, device="cuda")
first = None
for group in sys.argv[1:] or [""]:
    saved = {}
    for kv in filter(None, group.split(",")):
        k, v = kv.split("=")
        saved[k] = ctx.get_option(k)
        ctx.set_option(k, int(v))
    for _ in range(5):
        ctx.pipeline_pairs_dev(imgs.data_ptr(), 1, 480, 640, 640, 480 * 640, 2000, 0.75, synth.TUM_K, 1.0, 0.999, 0, res.data_ptr())
        ctx.sync()
    out = res.cpu().numpy().tobytes()
    first = first or out
    n = 300
    t0 = time.perf_counter()
    for _ in range(n):
        ctx.pipeline_pairs_dev(imgs.data_ptr(), 1, 480, 640, 640, 480 * 640, 2000, 0.75, synth.TUM_K, 1.0, 0.999, 0, res.data_ptr())
        ctx.sync()
    print(f"[{group or 'defaults'}] {1e3 * (time.perf_counter() - t0) / n:.4f} ms per fenced call; records equal to the first group's: {out == first}", flush=True)
    for k, v in saved.items():
        ctx.set_option(k, v)
