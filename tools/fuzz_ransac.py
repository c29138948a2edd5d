#!/usr/bin/env python3
"""Random two-view correspondence sets through vo_find_essential / vo_estimate_motion and the CPU
oracle: identical sample sets, iteration counts, inlier counts and masks are required, E to 1e-8,
R and t closely.  tools/fuzz_ransac.py [--cases N] [--seed S]
Scenes: random 3-D points in front of two cameras (random small or large baseline, rotation up
to 20 degrees, sometimes planar, sometimes pure rotation), pixel noise 0..2 px, 0..90 % gross
outliers, 5..3000 matches, thresholds 0.3..3 px, several intrinsics.
Inputs that are not well-posed - pure rotation (every translation fits), a baseline of a
five-hundredth of the depth (hardly better), sets in which a third of the matches are copies of one
(most minimal samples then hold a repeated point) - leave the five-point problem rank-deficient or
nearly so: its solutions come in clusters and families, which of them a solver reports is a matter of
its root finder and its gates, and the two sides agree on most such cases but not on all.  They are
run and listed ("degn" where the sides differ) but not counted.  On a well-posed scene a difference
must be explained by a minimal sample on which the two five-point stages report different model
sets ("incm": both approximate the complete solution set, see tools/dbg_model_sets.py); anything
else is a failure."""
import argparse, os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: F401  (one HIP runtime for both)
from visual_odometry_amd import capi, synth
from oracle import oracle as O

ap = argparse.ArgumentParser()
ap.add_argument("--cases", type=int, default=100)
ap.add_argument("--seed", type=int, default=1)
args = ap.parse_args()
ctx = capi.Context(0)
def one_sign(E):
    out = []
    for e in np.asarray(E).reshape(-1, 9):
        big = np.flatnonzero(np.abs(e) > 0.5 * np.max(np.abs(e)))[0]
        out.append(-e if e[big] < 0 else e)
    return np.array(out).reshape(-1, 9)


bad = 0
for case in range(args.cases):
    rng = np.random.default_rng(1000003 * args.seed + case)
    p1, p2, K, thr, fixed, info = synth.two_view_scene(rng)
    og, oE, om, oit, osamp = O.find_essential(p1, p2, K, threshold=thr, fixed_iters=fixed)
    gg, gE, gm, git, gsamp = ctx.find_essential(p1, p2, K, threshold=thr, fixed_iters=fixed)
    why = []
    if git != oit:
        why.append(f"iterations {git} vs {oit}")
    elif not np.array_equal(gsamp, osamp):
        why.append("sample sets")
    if gg != og or not np.array_equal(gm, om):
        why.append(f"inliers {gg} vs {og}, masks differ in {int(np.sum(gm != om))}")
    if og > 0:
        s = 1.0 if np.sum(gE * oE) > 0 else -1.0
        if np.max(np.abs(gE - s * oE)) > 1e-8:
            why.append(f"E differs by {np.max(np.abs(gE - s * oE)):.2e}")
    om_ = O.estimate_motion(p1, p2, K, threshold=thr, fixed_iters=fixed)
    gm_ = ctx.estimate_motion(p1, p2, K, threshold=thr, fixed_iters=fixed)
    if bool(om_["valid"]) != bool(gm_["valid"]) or om_["inliers"] != gm_["inliers"]:
        why.append(f"motion valid/inliers {gm_['valid']}/{gm_['inliers']} vs {om_['valid']}/{om_['inliers']}")
    elif om_["valid"]:
        # (the tolerances of the parity tests: a small baseline amplifies the 1e-9 between the two E)
        dR = np.max(np.abs(np.asarray(om_["R"]).reshape(3, 3) - np.asarray(gm_["R"]).reshape(3, 3)))
        dt = np.max(np.abs(np.asarray(om_["t"]).ravel() - np.asarray(gm_["t"]).ravel()))
        if dR > 1.7e-4 or dt > 1e-4:
            why.append(f"pose: R {dR:.1e} t {dt:.1e}")
    # not well-posed: repeated points, no translation, or a baseline of a five-hundredth of the depth
    degenerate = info[0] in (3, 4) or info[2] < 0.05
    incomplete = False
    if why and not degenerate and len(p1) > 5:
        # do the two five-point stages report the same model sets on the minimal samples both walked?
        k = min(git, oit)
        fx, fy, cx, cy = K
        n1 = np.stack([(p1[:, 0].astype(np.float64) - cx) / fx, (p1[:, 1].astype(np.float64) - cy) / fy], 1)
        n2 = np.stack([(p2[:, 0].astype(np.float64) - cx) / fx, (p2[:, 1].astype(np.float64) - cy) / fy], 1)
        gpu_models = ctx.five_point(n1[osamp[:k]], n2[osamp[:k]])
        for h in range(k):
            A, B = one_sign(gpu_models[h]), one_sign(O.five_point(n1[osamp[h]], n2[osamp[h]]))
            if len(A) != len(B) or (len(A) and max(np.min(np.max(np.abs(A - e), 1)) for e in B) > 1e-7):
                incomplete = True
                why.append(f"model sets differ at sample {h}: {len(A)} vs {len(B)}")
                break
    tag = "ok  " if not why else ("degn" if degenerate else "incm" if incomplete else "FAIL")
    bad += bool(why) and not degenerate and not incomplete
    print(f"{tag} case {case}: kind {info[0]} noise {info[1]} baseline {info[2]:.2f} n {len(p1)} thr {thr} fixed {fixed} iterations {git} inliers {gg}" + ("  " + "; ".join(why) if why else ""),
          flush=True)
print(f"failures: {bad}")
sys.exit(1 if bad else 0)
