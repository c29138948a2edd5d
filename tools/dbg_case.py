#!/usr/bin/env python3
"""One case of tests/test_gpu_parity.py::test_random_two_view_scenes taken apart: which hypothesis'
model sets differ between the GPU and the oracle.  tools/dbg_case.py <seed_mult> <case>"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: F401
from visual_odometry_amd import capi, synth
from oracle import oracle as O

mult, case = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(1000003 * mult + case)
p1, p2, K, thr, fixed, info = synth.two_view_scene(rng)
print("info", info, "n", len(p1), "thr", thr, "fixed", fixed, "K", K)
ctx = capi.Context(0)
og, oE, om, oit, osamp = O.find_essential(p1, p2, K, threshold=thr, fixed_iters=fixed)
gg, gE, gm, git, gsamp = ctx.find_essential(p1, p2, K, threshold=thr, fixed_iters=fixed)
print("oracle", og, oit, "gpu", gg, git, "samples equal", np.array_equal(gsamp, osamp))
fx, fy, cx, cy = K
n1 = np.stack([(p1[:, 0].astype(np.float64) - cx) / fx, (p1[:, 1].astype(np.float64) - cy) / fy], 1)
n2 = np.stack([(p2[:, 0].astype(np.float64) - cx) / fx, (p2[:, 1].astype(np.float64) - cy) / fy], 1)
x1 = n1[osamp]
x2 = n2[osamp]
gms = ctx.five_point(x1, x2)
bad = 0
for h in range(len(osamp)):
    oms = O.five_point(x1[h], x2[h])
    a = gms[h]
    def norm(E):
        E = E.reshape(-1, 9)
        out = []
        for e in E:
            s = 1.0
            m = np.max(np.abs(e))
            for v in e:
                if abs(v) > 0.5 * m:
                    s = -1.0 if v < 0 else 1.0
                    break
            out.append(s * e)
        return np.array(out).reshape(-1, 9)
    A, B = norm(np.asarray(a)), norm(np.asarray(oms))
    miss_g = [j for j, e in enumerate(B) if not len(A) or np.min(np.max(np.abs(A - e), 1)) > 1e-7]
    miss_o = [j for j, e in enumerate(A) if not len(B) or np.min(np.max(np.abs(B - e), 1)) > 1e-7]
    if miss_g or miss_o:
        bad += 1
        if bad <= 6:
            print(f"hypothesis {h} sample {osamp[h]}: gpu {len(A)} models, oracle {len(B)}; only oracle {miss_g}, only gpu {miss_o}")
            for j in miss_g:
                print("   oracle-only", np.array2string(B[j], precision=6))
            for j in miss_o:
                print("   gpu-only   ", np.array2string(A[j], precision=6))
print("hypotheses with different model sets:", bad, "of", len(osamp))
