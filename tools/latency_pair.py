#!/usr/bin/env python3
"""Latency of the per-pair host calls a caller of the reference's loop makes (one match_images and
one estimate_motion per pair, ref:src/main.cpp:152-184): milliseconds per call on one synthetic pair."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: F401  (one HIP runtime for both)
from visual_odometry_amd import capi, synth

ctx = capi.Context(0)
a, b = synth.frame_pair(3)
K = synth.default_K(a.shape[1], a.shape[0])
for name, (x, y) in {"dense": (a, b), "sparse": synth.frame_pair(3, sparse=True, moving_fraction=0.5)}.items():
    k1, k2, qi, ti, d = ctx.match_images(x, y, 2000, 0.75)
    p1 = np.stack([k1["x"][qi], k1["y"][qi]], 1)
    p2 = np.stack([k2["x"][ti], k2["y"][ti]], 1)
    for _ in range(5):
        ctx.match_images(x, y, 2000, 0.75)
        ctx.estimate_motion(p1, p2, K)
    n = 200
    t0 = time.perf_counter()
    for _ in range(n):
        ctx.match_images(x, y, 2000, 0.75)
    t1 = time.perf_counter()
    for _ in range(n):
        r = ctx.estimate_motion(p1, p2, K)
    t2 = time.perf_counter()
    print(f"{name}: {len(qi)} matches; match_images {1e3 * (t1 - t0) / n:.3f} ms, estimate_motion {1e3 * (t2 - t1) / n:.3f} ms "
          f"(inliers {r['inliers']})", flush=True)
