#!/usr/bin/env python3
"""Randomised parity sweep of the GPU detector, matcher and pipeline against the CPU oracle
(a one-off check on a GPU box; the fixed cases live in tests/).

    python tools/fuzz_parity.py [--cases 24] [--seed 1]

With VO_FAST_RAISE=1 in the environment the detector runs with raised FAST thresholds on these
single images as well; the last line then says on how many levels the threshold was raised and
how many had to run again.  Every third case is a low-texture scene, every fifth has a band of
strong texture on weak ground (a sample of tiles that over- or under-promises).
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

ap = argparse.ArgumentParser()
ap.add_argument("--cases", type=int, default=24)
ap.add_argument("--seed", type=int, default=1)
args = ap.parse_args()

import torch  # noqa: E402,F401  (before the HIP library: one HIP runtime for both)
from oracle import oracle as O  # noqa: E402
from visual_odometry_amd import capi, synth  # noqa: E402

O.lib()
ctx = capi.Context(0)
rng = np.random.default_rng(args.seed)
bad = raised_levels = redone_levels = 0
for case in range(args.cases):
    w = int(rng.integers(70, 1300))
    h = int(rng.integers(70, 800))
    nfeat = int(rng.choice([50, 300, 1000, 2000, 4000]))
    a, b = synth.frame_pair(1000 + case, height=h, width=w, sparse=(case % 3 == 1))
    img = a if case % 2 == 0 else b
    if case % 5 == 2:
        img = (100 + (img.astype(np.int32) - 128) // 8).astype(np.uint8)   # weak contrast everywhere ...
        y0, y1 = sorted(int(v) for v in rng.integers(0, h, 2))
        img[y0:y1] = (a if case % 2 == 0 else b)[y0:y1]                       # ... but for a band of rows
    if case % 3 == 0:   # strided view: rows padded by an odd amount
        pad = int(rng.integers(1, 40))
        big = np.zeros((h, w + pad), np.uint8)
        big[:, :w] = img
        img = big[:, :w]
    gk, gd = ctx.orb_detect(img, nfeat)
    info = ctx.fast_thresholds(0)
    if info is not None:
        raised_levels += int(np.count_nonzero(info[0] > 20))
        redone_levels += int(info[1].sum())
    ok, od = O.orb_detect(np.ascontiguousarray(img), nfeat)
    same = len(gk) == len(ok) and gk.tobytes() == ok.tobytes() and np.array_equal(gd, od)
    msg = f"case {case}: {w}x{h} nfeat {nfeat} contiguous={img.flags['C_CONTIGUOUS']} keypoints {len(gk)} / {len(ok)}"
    if len(ok) >= 2 and same:
        half = len(ok) // 2
        gi = ctx.match_ratio(gd[:half], gd[half:], 0.8)
        oi = O.match(od[:half], od[half:], 0.8)
        same = all(np.array_equal(x, y) for x, y in zip(gi, oi))
        msg += f" matches {len(oi[0])}"
    print(("ok   " if same else "FAIL ") + msg, flush=True)
    bad += not same
print("failures:", bad, "| levels with a raised FAST threshold:", raised_levels, "of which run again:", redone_levels)
sys.exit(1 if bad else 0)
