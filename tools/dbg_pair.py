"""Debug aid: one pair of the synthetic sequence, RANSAC stage on the GPU against the oracle."""
import sys
import numpy as np
sys.path.insert(0, ".")
import torch  # noqa
from visual_odometry_amd import capi, synth
from oracle import oracle as O

p = int(sys.argv[1]) if len(sys.argv) > 1 else 55
frames = synth.sequence(0x5EED0000, p + 2)
K = synth.TUM_K
ka, da = O.orb_detect(frames[p], 1000)
kb, db = O.orb_detect(frames[p + 1], 1000)
qi, ti, _ = O.match(da, db)
p1 = np.stack([ka["x"][qi], ka["y"][qi]], 1)
p2 = np.stack([kb["x"][ti], kb["y"][ti]], 1)
np.savez("gpurun_out/dbg_pair_pts.npz", p1=p1, p2=p2)
ctx = capi.Context(0)
og, oE, om, oit, osamp = O.find_essential(p1, p2, K)
gg, gE, gm, git, gsamp = ctx.find_essential(p1, p2, K)
print("n", len(p1), "oracle good/iters", og, oit, "gpu good/iters", gg, git)
print("samples equal over common prefix", np.array_equal(gsamp[:min(git, oit)], osamp[:min(git, oit)]))
s = 1.0 if np.sum(gE * oE) > 0 else -1.0
print("E diff", np.max(np.abs(gE - s * oE)), "mask equal", np.array_equal(gm, om))
o = O.estimate_motion(p1, p2, K)
g = ctx.estimate_motion(p1, p2, K)
print("oracle pose", o["valid"], o["inliers"], o["t"])
print("gpu pose   ", g["valid"], g["inliers"], g["t"])
# pose recovery of the oracle's E through both
cnt, R, t, m = O.recover_pose(oE, p1, p2, K, om)
print("oracle recover_pose on oracle E:", cnt, t)
cnt, R, t, m = O.recover_pose(gE, p1, p2, K, gm)
print("oracle recover_pose on gpu E:", cnt, t)

# per-hypothesis comparison of the minimal solver
fx, fy, cx, cy = K
x1 = np.stack([(p1[:, 0].astype(np.float64) - cx) / fx, (p1[:, 1].astype(np.float64) - cy) / fy], 1)
x2 = np.stack([(p2[:, 0].astype(np.float64) - cx) / fx, (p2[:, 1].astype(np.float64) - cy) / fy], 1)
thr = 1.0 / ((fx + fy) / 2)


def count(E):
    X1 = np.concatenate([x1, np.ones((len(x1), 1))], 1)
    X2 = np.concatenate([x2, np.ones((len(x2), 1))], 1)
    Ex1 = X1 @ E.T
    Etx2 = X2 @ E
    num = np.sum(X2 * Ex1, 1) ** 2
    den = Ex1[:, 0] ** 2 + Ex1[:, 1] ** 2 + Etx2[:, 0] ** 2 + Etx2[:, 1] ** 2
    return int(np.sum((num / den).astype(np.float32) <= np.float32(thr * thr)))


S = osamp[:oit]
got = ctx.five_point(x1[S], x2[S])
for h in range(len(S)):
    ref = O.five_point(x1[S[h]], x2[S[h]])
    gc = sorted(count(E) for E in got[h])
    oc = sorted(count(E) for E in ref)
    flag = "" if gc == oc else "   <-- differs"
    print(h, "gpu", len(got[h]), gc, "oracle", len(ref), oc, flag)
