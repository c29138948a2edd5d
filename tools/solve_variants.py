"""A/B of the five-point root kernels (VO_OPT_SOLVE_VARIANT) on the benchmark's sparse scene:
single-stream time of the solve group per variant, and whether the records agree."""
import os
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from visual_odometry_amd import capi, synth

pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 256
variants = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0, 1]
frames = []
for k in range(pairs):
    frames.extend(synth.frame_pair(0x5EED0000 + 2000 + k, sparse=True, moving_fraction=0.6, n_layers=24))
imgs = torch.from_numpy(np.stack(frames)).cuda()
ctx = capi.Context(0)
res = torch.zeros(pairs * capi.PAIR_DTYPE.itemsize, dtype=torch.uint8, device="cuda")
K = synth.TUM_K
ref = None
for v in variants:
    ctx.set_option("solve_variant", v)
    for rep in range(2):
        ctx.pipeline_pairs_dev(imgs.data_ptr(), pairs, 480, 640, 640, 480 * 640, 2000, 0.75, K, 1.0, 0.999, 0, res.data_ptr())
        ctx.sync()
    out = res.cpu().numpy().tobytes()
    if ref is None:
        ref = out
    elif out != ref:
        a, b = np.frombuffer(ref, capi.PAIR_DTYPE), np.frombuffer(out, capi.PAIR_DTYPE)
        for name in capi.PAIR_DTYPE.names:
            bad = np.flatnonzero([not np.array_equal(x, y) for x, y in zip(a[name], b[name])])
            if len(bad):
                print("   field", name, "differs on", len(bad), "pairs; first", bad[0], a[name][bad[0]], b[name][bad[0]])
    ctx.set_timing(True, ["ransac_solve", "ransac_score"])
    for rep in range(3):
        ctx.pipeline_pairs_dev(imgs.data_ptr(), pairs, 480, 640, 640, 480 * 640, 2000, 0.75, K, 1.0, 0.999, 0, res.data_ptr())
        ctx.sync()
    ms, n = ctx.kernel_time_ms(capi.T_GROUPS.index("ransac_solve"))
    ctx.set_timing(False)
    print(f"variant {v:2d}: solve {ms * n / 3:.3f} ms per call of {pairs} pairs; records equal: {out == ref}", flush=True)
