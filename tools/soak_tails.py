#!/usr/bin/env python3
"""Soak of the pipeline's overlapping tails (a one-off check on a GPU box): hundreds of asynchronous
calls over batches of different sizes and scenes, three record buffers in rotation, every call's
records read back after vo_pipeline_wait(ticket, results) while younger calls are in flight and
compared with the records the same batch gives when run alone.

    python tools/soak_tails.py [--calls 600] [--seed 1]
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument("--calls", type=int, default=600)
ap.add_argument("--seed", type=int, default=1)
ap.add_argument("--sizes", default="1,2,3,5,8,12",
                help="batch sizes to draw from (between 64 and 256 pairs of 2000 keypoints the kNN search runs on the tail stream too)")
args = ap.parse_args()

import torch  # noqa: E402
from visual_odometry_amd import capi, synth  # noqa: E402

ctx = capi.Context(0)
K = synth.TUM_K
rng = np.random.default_rng(args.seed)
batches = []
for b in range(7):
    n = int(rng.choice([int(v) for v in args.sizes.split(",")]))
    sparse = b % 3 == 1
    frames = []
    for p in range(n):
        frames += list(synth.frame_pair(900 + 20 * b + p, sparse=True, moving_fraction=0.6, n_layers=24) if sparse
                       else synth.frame_pair(900 + 20 * b + p))
    batches.append((n, torch.from_numpy(np.stack(frames)).cuda()))
NMAX = max(n for n, _ in batches)


def launch(b, res):
    n, imgs = batches[b]
    ctx.pipeline_pairs_dev(imgs.data_ptr(), n, 480, 640, 640, 480 * 640, 2000, 0.75, K, 1.0, 0.999, 0, res.data_ptr())
    return ctx.pipeline_ticket()


def records(res, n):
    return np.frombuffer(res.cpu().numpy().tobytes(), capi.PAIR_DTYPE)[:n].copy()


alone = []
for b, (n, _) in enumerate(batches):
    res = torch.zeros(NMAX * capi.PAIR_DTYPE.itemsize, dtype=torch.uint8, device="cuda")
    launch(b, res)
    ctx.sync()
    alone.append(records(res, n))
    print("batch", b, "pairs", n, "mean RANSAC iterations", float(alone[-1]["ransac_iters"].mean()), flush=True)

bufs = [torch.zeros(NMAX * capi.PAIR_DTYPE.itemsize, dtype=torch.uint8, device="cuda") for _ in range(3)]
inflight = []   # (ticket, batch, buffer)
bad = 0
for k in range(args.calls):
    b = int(rng.integers(0, len(batches)))
    if len(inflight) == len(bufs):          # the oldest call's buffer is needed: wait for that call only
        t, ob, obuf = inflight.pop(0)
        ctx.pipeline_wait(t, results=True)
        got = records(bufs[obuf], batches[ob][0])
        if got.tobytes() != alone[ob].tobytes():
            bad += 1
            print("MISMATCH call ticket", t, "batch", ob, flush=True)
        free = obuf
    else:
        free = len(inflight)
    inflight.append((launch(b, bufs[free]), b, free))
    if k % 100 == 99:
        print("calls", k + 1, "mismatches", bad, flush=True)
ctx.sync()
for t, ob, obuf in inflight:
    if records(bufs[obuf], batches[ob][0]).tobytes() != alone[ob].tobytes():
        bad += 1
print("mismatches:", bad)
sys.exit(1 if bad else 0)
