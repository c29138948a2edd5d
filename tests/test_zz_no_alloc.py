"""vo_match_ratio_static (the entry point that allocates nothing: its kNN(2) table lives in a
static device array of the library) against vo_match_ratio, the existing entry point it is a
variant of.  Bar: bit-exact query indices, train indices, distances and count.  Two calls in a
row on one context, the second with a smaller input, so that rows the first call left in the
shared workspace must not leak into the second result."""
import numpy as np
import pytest

from visual_odometry_amd import capi, synth

pytestmark = pytest.mark.gpu


def _noisy_copy(q, seed, flips=40, p=0.7):
    rng = np.random.default_rng(seed)
    t = q.copy()
    bits = rng.integers(0, 256, size=(len(t), flips))
    for k in range(flips):
        sel = rng.random(len(t)) < p
        t[sel, bits[sel, k] // 8] ^= (1 << (bits[sel, k] % 8)).astype(np.uint8)
    return t[rng.permutation(len(t))]


def _static(ctx, q, t, ratio):
    import torch
    nq, nt = len(q), len(t)
    dq = torch.from_numpy(q).cuda() if nq else torch.zeros(32, dtype=torch.uint8, device="cuda")
    dt = torch.from_numpy(t).cuda() if nt else torch.zeros(32, dtype=torch.uint8, device="cuda")
    qi = torch.full((max(nq, 1),), -5, dtype=torch.int32, device="cuda")
    ti = torch.full((max(nq, 1),), -5, dtype=torch.int32, device="cuda")
    di = torch.full((max(nq, 1),), -5.0, dtype=torch.float32, device="cuda")
    n = torch.full((1,), -5, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    ctx.match_ratio_static(dq.data_ptr(), nq, dt.data_ptr(), nt, ratio, qi.data_ptr(), ti.data_ptr(), di.data_ptr(),
                           n.data_ptr())
    ctx.sync()
    m = int(n.item())
    return qi[:m].cpu().numpy(), ti[:m].cpu().numpy(), di[:m].cpu().numpy(), (qi, ti, di)


def test_static_workspace_matcher_equals_match_ratio(ctx):
    q_big = synth.descriptors(901, 3000, dup_fraction=0.05)
    t_big = _noisy_copy(q_big, 902)
    q_small = synth.descriptors(903, 700, dup_fraction=0.1)
    t_small = _noisy_copy(q_small, 904)[:650]
    for q, t, ratio in [(q_big, t_big, 0.75), (q_small, t_small, 0.8)]:       # second call: smaller input
        want = ctx.match_ratio(q, t, ratio)
        got = _static(ctx, q, t, ratio)
        assert len(want[0]) > 100
        assert len(got[0]) == len(want[0])
        assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])
        assert np.array_equal(got[2], want[2])
        # nothing is written beyond the count
        m = len(want[0])
        assert (got[3][0][m:] == -5).all() and (got[3][1][m:] == -5).all()


def test_static_workspace_matcher_edges(ctx):
    q = synth.descriptors(905, 130)
    one = synth.descriptors(906, 1)
    empty = np.zeros((0, 32), np.uint8)
    assert len(_static(ctx, empty, q, 0.75)[0]) == 0          # no queries
    assert len(_static(ctx, q, empty, 0.75)[0]) == 0          # no train rows
    assert len(_static(ctx, q, one, 0.75)[0]) == 0            # a single neighbour never passes the ratio test
    want = ctx.match_ratio(q, q[::-1].copy(), 0.75)           # exact duplicates: distance 0 beats any second best
    got = _static(ctx, q, q[::-1].copy(), 0.75)
    assert len(want[0]) > 0
    assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1]) and np.array_equal(got[2], want[2])


def test_static_workspace_matcher_rejects_oversize(ctx):
    import torch
    buf = torch.zeros(64, dtype=torch.uint8, device="cuda")
    with pytest.raises(capi.VoError, match="exceed the static workspace"):
        ctx.match_ratio_static(buf.data_ptr(), 32768 + 1, buf.data_ptr(), 2, 0.75, buf.data_ptr(), buf.data_ptr(),
                               buf.data_ptr(), buf.data_ptr())
