#!/usr/bin/env python3
"""How often the GPU's five-point stage (first chart, second chart for odd model counts) reports a
different model set than the oracle's union over four charts, on minimal samples drawn from random
well-posed two-view scenes: tools/dbg_model_sets.py [scenes] [samples per scene]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: F401
from visual_odometry_amd import capi, synth
from oracle import oracle as O

n_scenes = int(sys.argv[1]) if len(sys.argv) > 1 else 30
per = int(sys.argv[2]) if len(sys.argv) > 2 else 500
ctx = capi.Context(0)


def norm(E):
    out = []
    for e in np.asarray(E).reshape(-1, 9):
        m = np.max(np.abs(e))
        s = 1.0
        for v in e:
            if abs(v) > 0.5 * m:
                s = -1.0 if v < 0 else 1.0
                break
        out.append(s * e)
    return np.array(out).reshape(-1, 9)


tot = miss_even = miss_odd = extra = 0
hist = {}
seen = 0
case = 0
while seen < n_scenes:
    rng = np.random.default_rng(1000003 * 11 + case)
    case += 1
    p1, p2, K, thr, fixed, info = synth.two_view_scene(rng)
    if info[0] in (3, 4) or info[2] < 0.05 or len(p1) < 20:
        continue
    seen += 1
    fx, fy, cx, cy = K
    n1 = np.stack([(p1[:, 0].astype(np.float64) - cx) / fx, (p1[:, 1].astype(np.float64) - cy) / fy], 1)
    n2 = np.stack([(p2[:, 0].astype(np.float64) - cx) / fx, (p2[:, 1].astype(np.float64) - cy) / fy], 1)
    idx = np.array([rng.choice(len(p1), 5, replace=False) for _ in range(per)])
    x1, x2 = n1[idx], n2[idx]
    gms = ctx.five_point(x1, x2)
    for h in range(per):
        A = norm(gms[h])
        O.set_five_point_charts(4)
        B = norm(O.five_point(x1[h], x2[h]))
        tot += 1
        mg = [j for j, e in enumerate(B) if not len(A) or np.min(np.max(np.abs(A - e), 1)) > 1e-7]
        mo = [j for j, e in enumerate(A) if not len(B) or np.min(np.max(np.abs(B - e), 1)) > 1e-7]
        if mo:
            extra += 1
        if mg:
            if len(A) % 2:
                miss_odd += 1
            else:
                miss_even += 1
            # how many oracle charts does it take to find what the GPU misses?
            need = 4
            for c in (1, 2, 3):
                O.set_five_point_charts(c)
                Bc = norm(O.five_point(x1[h], x2[h]))
                if len(Bc) == len(B):
                    need = c
                    break
            hist[(len(A), len(B), need)] = hist.get((len(A), len(B), need), 0) + 1
O.set_five_point_charts(4)
print(f"{tot} minimal samples: GPU misses models of the oracle on {miss_odd + miss_even} ({miss_odd} with an odd GPU count, "
      f"{miss_even} with an even one); GPU has models the oracle lacks on {extra}")
for k, v in sorted(hist.items()):
    print(f"   gpu {k[0]} models, oracle {k[1]}, oracle charts needed {k[2]}: {v}")
