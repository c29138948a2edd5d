#!/usr/bin/env python3
"""What the raised FAST thresholds look like on the benchmark scenes (a one-off check on a GPU box):
per level the threshold, the redo flags, and the share of the plain run's candidates left.

    python tools/fast_raise_probe.py [--sparse]
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument("--sparse", action="store_true")
ap.add_argument("--images", type=int, default=8)
args = ap.parse_args()

import torch  # noqa: E402,F401
from visual_odometry_amd import capi, synth  # noqa: E402

ctx = capi.Context(0)
for k in range(args.images):
    img = synth.frame_pair(500 + k, sparse=args.sparse)[0] if args.sparse else synth.frame_pair(500 + k)[0]
    ctx.set_option("fast_raise", 1)
    ctx.orb_detect(img, 2000)
    thr, redo = ctx.fast_thresholds(0)
    kept = [len(ctx.debug_candidates(0, l)) for l in range(8)]
    ctx.set_option("fast_raise", 0)
    ctx.orb_detect(img, 2000)
    full = [len(ctx.debug_candidates(0, l)) for l in range(8)]
    print("image", k, "thr", thr.tolist(), "redo", redo.tolist(), "candidates", kept, "of", full, flush=True)
