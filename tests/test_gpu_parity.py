"""GPU parity suite: every stage of the HIP path, called through the C ABI,
against the CPU oracle on the same seeded inputs.  Integer / byte / index
results must be bit-exact; pose tolerances are written in the tests."""
import os

import numpy as np
import pytest

from visual_odometry_amd import synth

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
KITTI_K = synth.KITTI_K
TUM_K = synth.TUM_K


def _rot_err_deg(R1, R2):
    c = (np.trace(R1.T @ R2) - 1) / 2
    return np.degrees(np.arccos(np.clip(c, -1, 1)))


# ------------------------------------------------------------------ Hamming kNN(2)
@pytest.mark.parametrize("nq,nt", [(1, 1), (1, 2), (5, 1), (64, 64), (777, 1234), (2000, 2000), (300, 257),
                                   (4100, 513)])
def test_knn2_matches_oracle(ctx, oracle, nq, nt):
    q = synth.descriptors(100 + nq, nq, dup_fraction=0.2)
    t = synth.descriptors(200 + nt, nt, dup_fraction=0.3)
    if nt > 100 and nq > 50:
        t[-50:] = q[:50]                      # exact duplicates: distance-0 ties
        t[7:57] = q[:50]
    gi, gd = ctx.hamming_knn2(q, t)
    oi, od = oracle.knn2(q, t)
    assert np.array_equal(gi, oi)
    assert np.array_equal(gd, od)


@pytest.mark.parametrize("knob", ["knn_waves=1", "knn_waves=4", "knn_waves=8", "knn_waves=16", "knn_mode=0", "knn_mode=1",
                                  "knn_mode=1,knn_waves=1", "knn_mode=1,knn_waves=4", "knn_mode=1,knn_waves=8",
                                  "knn_mode=1,knn_waves=16", "knn_queries_per_lane=2", "knn_queries_per_lane=4",
                                  "knn_train_split=2", "knn_train_split=5", "knn_train_split=64",
                                  "knn_mode=1,knn_train_split=5", "knn_mode=3", "knn_mode=3,knn_waves=1",
                                  "knn_mode=3,knn_waves=4", "knn_mode=3,knn_waves=8", "knn_mode=3,knn_train_split=5"])
@pytest.mark.parametrize("nq,nt", [(1, 1), (70, 3), (300, 257), (2000, 2000), (31, 4097), (1500, 70000)])
def test_knn2_every_launch_variant(ctx, oracle, options, knob, nq, nt):
    """The matcher picks its kernel and workgroup size from the problem size (the matrix-core kernel
    by default, mode 1 the vector-ALU one, one wavefront per workgroup only for batches of several
    hundred pairs, the scalar-operand kernels only beyond 65 536 train rows): every variant is
    forced here, on shapes with ragged last blocks, ties at distance 0, a scan long enough for the
    matrix-core kernel's test-before-update phase with a one-row last tile (31 x 4097 with one
    wavefront) and a train set longer than one chunk of either kernel."""
    for part in knob.split(","):
        name, value = part.split("=")
        options(name, int(value))
    q = synth.descriptors(300 + nq, nq, dup_fraction=0.2)
    t = synth.descriptors(400 + nt, nt, dup_fraction=0.3)
    if nt > 100 and nq > 50:
        t[-50:] = q[:50]
        t[7:57] = q[:50]
    gi, gd = ctx.hamming_knn2(q, t)
    oi, od = oracle.knn2(q, t)
    assert np.array_equal(gi, oi)
    assert np.array_equal(gd, od)


def test_knn2_random_shapes(ctx, oracle, options):
    """Forty searches of random small shapes (1 .. 400 query rows, 1 .. 6000 train rows, with exact
    copies and near copies planted) through the default kernel and through the matrix-core
    kernel's other forms: tables bit-identical to the oracle's, row for row."""
    rng = np.random.default_rng(2024)
    for case in range(40):
        nq = int(rng.integers(1, 401))
        nt = int(rng.integers(1, 6001)) if case % 4 else int(rng.integers(1, 40))
        q = rng.integers(0, 256, (nq, 32), dtype=np.uint8)
        t = rng.integers(0, 256, (nt, 32), dtype=np.uint8)
        for k in range(0, nq, 5):            # copies of train rows, some with one or two bits flipped
            q[k] = t[rng.integers(0, nt)]
            if k % 10 == 0:
                q[k, rng.integers(0, 32)] ^= np.uint8(1 << int(rng.integers(0, 8)))
        if nt > 3:
            t[nt - 1] = t[0]                 # a tie between the first and the last train row
            t[nt // 2] = t[0]
        oi, od = oracle.knn2(q, t)
        for knob in ("knn_mode=-1", "knn_mode=3", "knn_mode=2,knn_waves=1", "knn_mode=3,knn_waves=8,knn_train_split=3"):
            for part in knob.split(","):
                name, value = part.split("=")
                options(name, int(value))
            gi, gd = ctx.hamming_knn2(q, t)
            assert np.array_equal(gi, oi) and np.array_equal(gd, od), (case, nq, nt, knob)
            for part in knob.split(","):
                options(part.split("=")[0], -1 if part.startswith("knn_mode") else 0)


def test_knn2_empty_train(ctx):
    q = synth.descriptors(1, 10)
    gi, gd = ctx.hamming_knn2(q, np.zeros((0, 32), np.uint8))
    assert np.all(gi == -1) and np.all(gd == -1)


def test_knn2_beyond_65536_train_rows(ctx, oracle):
    q = synth.descriptors(31, 1500, dup_fraction=0.1)
    t = synth.descriptors(32, 70000, dup_fraction=0.05)
    t[69000:69100] = q[:100]
    t[100:200] = q[:100]
    gi, gd = ctx.hamming_knn2(q, t)
    oi, od = oracle.knn2(q, t)
    assert np.array_equal(gi, oi) and np.array_equal(gd, od)


def test_c4_full_size_65536x65536(ctx, oracle):
    """The 65 536 x 65 536 match of the benchmark at full size through vo_hamming_knn2_dev with
    the dispatch the library picks, once whole and once as two 32 768-row query shards (what
    two ranks would run).  More than 2 000 sampled query rows - the first and last row of 64-row
    blocks, the whole last block, rows with planted exact duplicates at the first, middle and
    last train rows - are checked against the oracle on the full train set; the two runs must
    agree on every row."""
    import torch
    nq = nt = 65536
    q = synth.descriptors(7, nq)
    t = synth.descriptors(8, nt)
    planted = np.array([0, 63, 64, 4097, 32767, 32768, 50000, 65472, 65535])
    t[0] = q[planted[0]]                       # a distance-0 pair at both ends of the train set: ties -> (0, 65535)
    t[65535] = q[planted[0]]
    t[32767] = q[planted[4]]
    t[32768] = q[planted[4]]
    for k, r in enumerate(planted[1:4]):
        t[1000 + k] = q[r]
    t[65534] = q[planted[8]]
    t[1] = q[planted[8]]
    dq, dt = torch.from_numpy(q).cuda(), torch.from_numpy(t).cuda()

    def run(rows):
        idx = torch.full((len(rows), 2), -7, dtype=torch.int32, device="cuda")
        dst = torch.full((len(rows), 2), -7, dtype=torch.int32, device="cuda")
        part = dq[rows[0]: rows[-1] + 1]
        ctx.hamming_knn2_dev(part.data_ptr(), len(rows), dt.data_ptr(), nt, idx.data_ptr(), dst.data_ptr())
        ctx.sync()
        return idx.cpu().numpy(), dst.cpu().numpy()

    gi, gd = run(np.arange(nq))
    a = run(np.arange(0, nq // 2))
    b = run(np.arange(nq // 2, nq))
    assert np.array_equal(np.concatenate([a[0], b[0]]), gi) and np.array_equal(np.concatenate([a[1], b[1]]), gd)
    rng = np.random.default_rng(3)
    blocks = rng.choice(nq // 64, 600, replace=False)
    rows = np.unique(np.concatenate([blocks * 64, blocks * 64 + 63, np.arange(nq - 64, nq), planted,
                                     rng.integers(0, nq, 900)]))
    assert len(rows) >= 2048
    oi, od = oracle.knn2(q[rows], t)
    assert np.array_equal(gi[rows], oi) and np.array_equal(gd[rows], od)
    assert list(gi[planted[0]]) == [0, 65535] and list(gd[planted[0]]) == [0, 0]
    assert list(gi[planted[4]]) == [32767, 32768]
    assert list(gi[planted[8]]) == [1, 65534]


def _c4_descriptors(nq, nt):
    q = synth.descriptors(7, nq)
    t = synth.descriptors(8, nt)
    for k in range(1, 64):                     # a distance-0 pair across every 64th of the train set
        t[k * (nt // 64) - 1] = q[64 + k]      # (the chunk boundaries of a split search)
        t[k * (nt // 64)] = q[64 + k]
    t[0] = q[0]                                # ... at both ends of the train set
    t[nt - 1] = q[0]
    t[nt // 2 - 1] = q[nq // 2]
    t[nt // 2] = q[nq // 2]
    t[nt - 2] = q[nq - 1]
    t[1] = q[nq - 1]
    return q, t


def test_c4_shard_of_8192_query_rows_is_split_over_the_train_set(ctx, oracle, options):
    """One rank's share of the 65 536 x 65 536 match: 8 192 query rows are 128 workgroups, too few
    for 256 compute units, so the library also splits the train set (grid = query blocks x chunks,
    per-chunk top two merged by key).  The result must equal the unsplit search bit for bit -
    checked on every row against a forced single-chunk run, and on sampled rows (tied pairs that
    straddle chunk boundaries among them) against the oracle."""
    import torch
    nq, nt = 8192, 65536
    q, t = _c4_descriptors(nq, nt)
    dq, dt = torch.from_numpy(q).cuda(), torch.from_numpy(t).cuda()

    def run():
        idx = torch.full((nq, 2), -7, dtype=torch.int32, device="cuda")
        dst = torch.full((nq, 2), -7, dtype=torch.int32, device="cuda")
        ctx.hamming_knn2_dev(dq.data_ptr(), nq, dt.data_ptr(), nt, idx.data_ptr(), dst.data_ptr())
        ctx.sync()
        return idx.cpu().numpy(), dst.cpu().numpy()

    gi, gd = run()                             # the library's own choice: split
    options("knn_train_split", 1)
    wi, wd = run()
    assert np.array_equal(gi, wi) and np.array_equal(gd, wd)
    for split in (3, 8, 64):
        options("knn_train_split", split)
        si, sd = run()
        assert np.array_equal(si, wi) and np.array_equal(sd, wd), split
    rows = np.unique(np.concatenate([np.arange(0, 192), np.arange(nq - 64, nq), [nq // 2],
                                     np.random.default_rng(5).integers(0, nq, 1000)]))
    oi, od = oracle.knn2(q[rows], t)
    assert np.array_equal(gi[rows], oi) and np.array_equal(gd[rows], od)
    assert list(gi[0]) == [0, nt - 1] and list(gd[0]) == [0, 0]
    assert list(gi[nq // 2]) == [nt // 2 - 1, nt // 2]
    assert list(gi[64 + 5]) == [5 * (nt // 64) - 1, 5 * (nt // 64)]


@pytest.mark.parametrize("n_ctx,nq,nt", [(2, 4096, 8192), (3, 1000, 3000), (4, 3, 70), (8, 16384, 16384)])
def test_knn2_over_several_contexts_equals_one(ctx, oracle, n_ctx, nq, nt):
    """vo_hamming_knn2_multi / _multi_dev: query rows in contiguous blocks over several contexts
    (here all on the one GPU of the box; on a node one per GPU), train set replicated, every block's
    rows sent to their place in the first context's tables.  Equal to one context's search."""
    import torch
    from visual_odometry_amd import capi
    q = synth.descriptors(900 + nq, nq, dup_fraction=0.2)
    t = synth.descriptors(901 + nt, nt, dup_fraction=0.2)
    if nt > 100 and nq > 50:
        t[-50:] = q[:50]
        t[7:57] = q[:50]
    wi, wd = ctx.hamming_knn2(q, t)
    oi, od = oracle.knn2(q[: min(nq, 1500)], t)
    assert np.array_equal(wi[: len(oi)], oi) and np.array_equal(wd[: len(od)], od)
    ctxs = [capi.Context(0) for _ in range(n_ctx)]
    try:
        gi, gd = capi.hamming_knn2_multi(ctxs, q, t)
        assert np.array_equal(gi, wi) and np.array_equal(gd, wd)
        dq, dt = torch.from_numpy(q).cuda(), torch.from_numpy(t).cuda()
        copies = [dt.clone() for _ in range(n_ctx)]                    # one replica per context
        blocks = [capi.block_range(nq, k, n_ctx) for k in range(n_ctx)]
        parts = [dq[b:e].clone() if e > b else dq[:1].clone() for b, e in blocks]
        idx = torch.full((nq, 2), -7, dtype=torch.int32, device="cuda")
        dst = torch.full((nq, 2), -7, dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()
        for rep in range(2):                                           # back to back: the per-context tables are reused
            capi.hamming_knn2_multi_dev(ctxs, [p.data_ptr() for p in parts], nq, [c.data_ptr() for c in copies], nt,
                                        idx.data_ptr(), dst.data_ptr())
        for c in ctxs:
            c.sync()
        assert np.array_equal(idx.cpu().numpy(), wi) and np.array_equal(dst.cpu().numpy(), wd)
    finally:
        for c in ctxs:
            c.close()


@pytest.mark.parametrize("ratio", [0.5, 0.75, 0.9])
def test_match_ratio_matches_oracle(ctx, oracle, ratio):
    q = synth.descriptors(41, 1800, dup_fraction=0.05)
    t = q.copy()
    rng = np.random.default_rng(42)
    flips = rng.integers(0, 256, size=(len(t), 40))
    for k in range(40):                       # noisy copies: many true matches, graded distances
        sel = rng.random(len(t)) < 0.7
        t[sel, flips[sel, k] // 8] ^= (1 << (flips[sel, k] % 8)).astype(np.uint8)
    t = t[rng.permutation(len(t))]
    gq, gt, gd = ctx.match_ratio(q, t, ratio)
    oq, ot, od = oracle.match(q, t, ratio)
    assert len(oq) > 100
    assert np.array_equal(gq, oq) and np.array_equal(gt, ot) and np.array_equal(gd, od)


def test_match_empty_descriptors(ctx):
    d = synth.descriptors(5, 20)
    e = np.zeros((0, 32), np.uint8)
    assert len(ctx.match_ratio(e, d)[0]) == 0
    assert len(ctx.match_ratio(d, e)[0]) == 0


# ------------------------------------------------------------------ ORB stages
IMAGES = {
    "synth640": lambda: synth.frame_pair(7)[0],
    "checker": lambda: synth.checkerboard(seed=9),
    "kitti": lambda: synth.frame_pair(8, height=376, width=1241)[0],
    "odd": lambda: synth.frame_pair(9, height=301, width=423)[1],
    # one blur / FAST tile column only, widths that are not multiples of four, a level pyramid whose
    # upper levels are smaller than the 31-px border
    "narrow": lambda: synth.frame_pair(10, height=203, width=75)[0],
    "low": lambda: synth.frame_pair(11, height=66, width=517)[1],
    "edge68": lambda: synth.frame_pair(12, height=117, width=131)[0],
    # grey levels 20 and 21 apart in one- to three-pixel blocks: ring-minus-centre differences sit on
    # both sides of the FAST threshold (20) and exactly on it, in both polarities
    "threshold": lambda: _threshold_steps(13, 240, 333),
}


def _threshold_steps(seed, height, width):
    rng = np.random.default_rng(seed)
    levels = np.array([70, 90, 91, 110, 111, 131, 132, 152], np.uint8)
    img = np.zeros((height, width), np.uint8)
    y = 0
    while y < height:
        bh = int(rng.integers(1, 4))
        x = 0
        while x < width:
            bw = int(rng.integers(1, 4))
            img[y:y + bh, x:x + bw] = levels[rng.integers(0, len(levels))]
            x += bw
        y += bh
    return img


@pytest.mark.parametrize("name", list(IMAGES))
def test_pyramid_blur_fast_stages(ctx, oracle, name):
    img = IMAGES[name]()
    ctx.orb_detect(img, 1000)
    for l in range(8):
        lv = ctx.debug_level(0, l)
        ol = oracle.pyramid_level(img, l)
        assert lv.shape == ol.shape
        assert np.array_equal(lv, ol), f"pyramid level {l}"
        assert np.array_equal(ctx.debug_level(0, l, blurred=True), oracle.pyramid_level(img, l, blurred=True)), \
            f"blurred level {l}"
        cand = ctx.debug_candidates(0, l)
        h, w = ol.shape
        of = oracle.fast(ol)
        keep = (of[:, 0] >= 31) & (of[:, 0] < w - 31) & (of[:, 1] >= 31) & (of[:, 1] < h - 31) if len(of) else []
        of = of[keep] if len(of) else of
        got = sorted(map(tuple, cand.tolist()))
        exp = sorted(map(tuple, of.tolist()))
        assert got == exp, f"FAST candidates level {l}: {len(got)} vs {len(exp)}"


def _strong_texture_on_sampled_tiles(seed, height=480, width=640):
    """Weak corners (grey steps of 26) everywhere, strong ones (steps of 190) only inside the level-0
    FAST tiles that are sampled at the plain threshold: the sample promises far more strong
    candidates than the level holds, so the raised threshold is too high and the level's other
    tiles have to run again."""
    rng = np.random.default_rng(seed)

    def blocks(lo, hi):
        img = np.zeros((height, width), np.uint8)
        y = 0
        while y < height:
            bh = int(rng.integers(2, 5))
            x = 0
            while x < width:
                bw = int(rng.integers(2, 5))
                img[y:y + bh, x:x + bw] = lo if rng.random() < 0.5 else hi
                x += bw
            y += bh
        return img

    img, strong = blocks(100, 126), blocks(30, 220)
    for ty in range((height - 62 + 31) // 32):
        for tx in range((width - 62 + 63) // 64):
            if ((tx >> 1) + 3 * ty) % 8 == 0:
                ys, xs = 31 + 32 * ty, 31 + 64 * tx
                img[ys:ys + 32, xs:xs + 64] = strong[ys:ys + 32, xs:xs + 64]
    return img


def test_orb_detect_when_the_pretest_queue_overflows(ctx, oracle):
    """FAST's first queue holds what half a tile can pass at most; a tile that passes more is worked
    through in two halves of its rows.  A diagonal sawtooth makes nearly every pixel pass the
    cardinal pre-test in both polarities (brighter to the north and east, darker to the south and
    west), which overflows the queue in every tile; noise on top of it leaves real corners.  The
    keypoints and descriptors must still be the oracle's, byte for byte."""
    rng = np.random.default_rng(31)
    yy, xx = np.mgrid[0:480, 0:640]
    saw = ((9 * (xx - yy)) % 256).astype(np.int32)
    for k, noise in enumerate((0, 12, 40)):
        img = np.clip(saw + rng.integers(-noise, noise + 1, saw.shape), 0, 255).astype(np.uint8)
        img[100:140, 200:260] = 0          # flat patches with sharp borders: some certain corners
        img[300:330, 400:470] = 255
        gk, gd = ctx.orb_detect(img, 1500)
        ok, od = oracle.orb_detect(img, 1500)
        assert len(gk) == len(ok) and len(gk) > 100, (k, len(gk), len(ok))
        assert gk.tobytes() == ok.tobytes(), k
        assert np.array_equal(gd, od), k



def test_orb_detect_batch_of_34_images_bit_exact(ctx, oracle):
    """A device-resident batch large enough for the forms the kernels take on large calls (raised FAST
    thresholds, sixteen keypoints per wavefront in the descriptor kernel - from 4096 wavefronts on, here
    34 images x 125): every image's keypoints and descriptors are the oracle's, byte for byte.  The
    images differ in content and in keypoint count (dense and sparse scenes, one nearly flat)."""
    import torch
    from visual_odometry_amd import capi
    frames = []
    for k in range(8):
        frames.extend(synth.frame_pair(4100 + k))
    for k in range(8):
        frames.extend(synth.frame_pair(0x5EED0000 + 2100 + k, sparse=True, moving_fraction=0.6, n_layers=24))
    flat = np.full((480, 640), 90, np.uint8)
    flat[200:280, 300:420] = 160                      # four corners (on several levels) and nothing else
    frames.extend([flat, frames[3][::-1].copy()])     # ... and an image turned upside down
    n_img, cap = len(frames), 2000 + 8 * 256
    assert n_img == 34
    imgs = torch.from_numpy(np.stack(frames)).cuda()
    kps = torch.zeros(n_img * cap * capi.KP_DTYPE.itemsize, dtype=torch.uint8, device="cuda")
    desc = torch.zeros(n_img * cap * 32, dtype=torch.uint8, device="cuda")
    cnt = torch.zeros(n_img, dtype=torch.int32, device="cuda")
    ctx.orb_detect_batch_dev(imgs.data_ptr(), n_img, 480, 640, 640, 480 * 640, 2000, kps.data_ptr(), desc.data_ptr(), cap,
                             cnt.data_ptr())
    ctx.sync()
    assert ctx.fast_thresholds(0) is not None          # the batch is large enough for raised thresholds
    counts = cnt.cpu().numpy()
    gk = np.frombuffer(kps.cpu().numpy().tobytes(), capi.KP_DTYPE).reshape(n_img, cap)
    gd = desc.cpu().numpy().reshape(n_img, cap, 32)
    seen = set()
    for i, img in enumerate(frames):
        ok, od = oracle.orb_detect(img, 2000)
        assert counts[i] == len(ok), i
        assert gk[i, :len(ok)].tobytes() == ok.tobytes(), i
        assert np.array_equal(gd[i, :len(ok)], od), i
        seen.add(len(ok))
    assert 2000 in seen and min(seen) < 64 and len(seen) >= 3   # full lists, a handful of keypoints, odd counts between


@pytest.mark.parametrize("name,nfeat", [("synth640", 2000), ("synth640", 300), ("checker", 2000), ("kitti", 4000),
                                        ("odd", 500), ("narrow", 300), ("threshold", 800), ("sampled_strong", 2000),
                                        ("sampled_strong", 500)])
def test_orb_detect_with_raised_fast_thresholds(ctx, oracle, options, name, nfeat):
    """Batched calls run most FAST tiles with a per-(image, level) threshold raised towards the
    score cut of the selection; forced here on single images.  The keypoints and descriptors must
    not change, the candidate lists may only lose candidates, and both outcomes - threshold kept,
    tiles run again - must occur."""
    img = _strong_texture_on_sampled_tiles(21) if name == "sampled_strong" else IMAGES[name]()
    options("fast_raise", 1)
    gk, gd = ctx.orb_detect(img, nfeat)
    info = ctx.fast_thresholds(0)
    assert info is not None
    thr, redo = info
    cands = [ctx.debug_candidates(0, l) for l in range(8)]
    options("fast_raise", 0)
    pk, pd = ctx.orb_detect(img, nfeat)
    assert ctx.fast_thresholds(0) is None
    ok, od = oracle.orb_detect(img, nfeat)
    assert len(ok) > 0
    assert gk.tobytes() == ok.tobytes() and np.array_equal(gd, od)
    assert pk.tobytes() == ok.tobytes() and np.array_equal(pd, od)
    assert np.all(thr % 2 == 0) and np.all(thr >= 20) and np.all(thr <= 240)
    for l in range(8):
        full = set(map(tuple, ctx.debug_candidates(0, l).tolist()))
        got = set(map(tuple, cands[l].tolist()))
        assert len(got) == len(cands[l]) and got <= full, f"level {l}"
        if thr[l] == 20 or redo[l]:
            assert got == full, f"level {l}: the list must be complete"
        else:   # everything at or above the raised threshold, plus the sampled tiles' weaker candidates
            assert {c for c in full if c[2] >= thr[l]} <= got, f"level {l}"
    if name == "synth640":
        assert thr[0] > 20 and not redo[0]
    if name == "sampled_strong":
        assert thr[0] > 20 and redo[0]


def test_pipeline_with_and_without_raised_fast_thresholds(ctx, options):
    """The batched pipeline raises the thresholds by itself from a dozen images on; its records
    must not depend on it."""
    import torch
    from visual_odometry_amd import capi
    frames = []
    for k in range(8):
        frames.extend(synth.frame_pair(300 + k))
    imgs = torch.from_numpy(np.stack(frames)).cuda()
    K = synth.default_K(640, 480)
    out = []
    for flag in (0, 1, None):
        options("fast_raise", -1 if flag is None else flag)
        res = torch.zeros(8 * capi.PAIR_DTYPE.itemsize, dtype=torch.uint8, device="cuda")
        ctx.pipeline_pairs_dev(imgs.data_ptr(), 8, 480, 640, 640, 480 * 640, 2000, 0.75, K, 1.0, 0.999, 0, res.data_ptr())
        ctx.sync()
        out.append(res.cpu().numpy().tobytes())
        if flag is None:
            assert ctx.fast_thresholds(3) is not None   # 16 images x 379 tiles: raised by default
    assert out[0] == out[1] == out[2]
    rec = np.frombuffer(out[0], capi.PAIR_DTYPE)
    assert np.all(rec["valid"] == 1) and np.all(rec["n_kp1"] == 2000)


@pytest.mark.parametrize("name,nfeat", [("synth640", 2000), ("synth640", 1000), ("checker", 2000), ("checker", 100),
                                        ("kitti", 4000), ("odd", 500), ("synth640", 37), ("narrow", 300),
                                        ("low", 300), ("edge68", 500), ("threshold", 800)])
def test_orb_detect_bit_exact(ctx, oracle, name, nfeat):
    img = IMAGES[name]()
    gk, gd = ctx.orb_detect(img, nfeat)
    ok, od = oracle.orb_detect(img, nfeat)
    assert len(gk) == len(ok) and len(gk) > 0
    for f in ("octave", "x", "y", "size", "response", "angle", "class_id"):
        assert np.array_equal(gk[f], ok[f]), f"keypoint field {f}"
    assert gk.tobytes() == ok.tobytes()
    assert np.array_equal(gd, od)
    gk2, none = ctx.orb_detect(img, nfeat, want_desc=False)     # detect_keypoints_only
    assert none is None and gk2.tobytes() == ok.tobytes()


def test_orb_strided_input(ctx, oracle):
    big = np.zeros((480, 700), np.uint8)
    img = synth.frame_pair(12)[0]
    big[:, :640] = img
    view = big[:, :640]                       # row step 700, not a multiple of 4 images wide
    import ctypes as C
    from visual_odometry_amd import capi
    kps = np.zeros(4096, capi.KP_DTYPE)
    desc = np.zeros((4096, 32), np.uint8)
    n = C.c_int()
    rc = ctx.lib.vo_orb_detect(ctx.h, big.ctypes.data_as(C.c_void_p), 480, 640, 700, 1000,
                               kps.ctypes.data_as(C.c_void_p), desc.ctypes.data_as(C.c_void_p), 4096, C.byref(n))
    assert rc == 0
    ok, od = oracle.orb_detect(np.ascontiguousarray(view), 1000)
    assert kps[: n.value].tobytes() == ok.tobytes() and np.array_equal(desc[: n.value], od)


def test_detector_reference_contract(ctx):
    # ref:tests/test_feature_detector.cpp:28-102
    img = synth.checkerboard(seed=1)
    k, d = ctx.orb_detect(img, 2000)
    assert len(k) > 0 and d.shape == (len(k), 32)
    assert len(ctx.orb_detect(img, 100, want_desc=False)[0]) <= 100
    assert len(ctx.orb_detect(np.zeros((0, 0), np.uint8))[0]) == 0
    assert len(ctx.orb_detect(np.full((480, 640), 128, np.uint8))[0]) < 10


def test_capacity_overflow_is_loud(ctx):
    from visual_odometry_amd import capi
    img = synth.frame_pair(7)[0]
    with pytest.raises(capi.VoError):
        ctx.orb_detect(img, 2000, cap=100)


def _dot_grid(spacing):
    """Isolated bright pixels on black: each is a FAST corner of score 254 with the same Harris
    response, so both retainBest cuts land inside one huge group of ties."""
    img = np.zeros((480, 640), np.uint8)
    img[4::spacing, 4::spacing] = 255
    return img


@pytest.mark.parametrize("spacing,bit,stage", [(8, 1, "FAST-score selection"), (12, 2, "Harris selection")])
def test_truncated_keypoint_lists_are_reported_on_the_device_resident_path(ctx, oracle, spacing, bit, stage):
    """OpenCV's retainBest keeps every tie of the last retained score, the GPU tables are finite:
    when a level's list does not fit, the call must not pass silently.  Device-resident calls only
    enqueue, so the report comes from vo_sync (once), the pair records carry the bit, and the cut
    list is the same on every run (first in raster order), not whichever lanes won a slot."""
    import torch
    from visual_odometry_amd import capi
    a = _dot_grid(spacing)
    ok, _ = oracle.orb_detect(a, 2000)
    per_level = np.bincount(ok["octave"], minlength=8)
    assert per_level[0] > (2048 if bit == 1 else 434 + 256), "the scene no longer overflows: pick another spacing"
    b = synth.frame_pair(7)[0]                                  # an ordinary image: its pairs stay clean
    imgs = torch.from_numpy(np.stack([a, b, b, b, b, a])).cuda()
    runs = []
    for rep in range(2):
        res = torch.zeros(3 * capi.PAIR_DTYPE.itemsize, dtype=torch.uint8, device="cuda")
        ctx.pipeline_pairs_dev(imgs.data_ptr(), 3, 480, 640, 640, 480 * 640, 2000, 0.75, TUM_K, 1.0, 0.999, 0,
                               res.data_ptr())
        with pytest.raises(capi.VoError, match=stage):
            ctx.sync()
        ctx.sync()                                              # reported once, then cleared
        out = np.frombuffer(res.cpu().numpy().tobytes(), capi.PAIR_DTYPE)
        # (a list cut at the score stage is still longer than a level's keypoint table: both bits)
        status = capi.TRUNC_SCORE_LIST | capi.TRUNC_KEYPOINT_LIST if bit == 1 else capi.TRUNC_KEYPOINT_LIST
        assert list(out["status"]) == [status, 0, status]
        runs.append(out.tobytes())
    assert runs[0] == runs[1]
    # the detector alone: same report, and a deterministic (raster-first) cut of the level-0 list
    kps = [torch.zeros(4096 * capi.KP_DTYPE.itemsize, dtype=torch.uint8, device="cuda") for _ in range(2)]
    cnt = torch.zeros(1, dtype=torch.int32, device="cuda")
    for k in kps:
        ctx.orb_detect_batch_dev(imgs.data_ptr(), 1, 480, 640, 640, 480 * 640, 2000, k.data_ptr(), None, 4096,
                                 cnt.data_ptr())
        with pytest.raises(capi.VoError, match=stage):
            ctx.sync()
    n = int(cnt.item())
    g = [np.frombuffer(k.cpu().numpy().tobytes(), capi.KP_DTYPE)[:n] for k in kps]
    assert g[0].tobytes() == g[1].tobytes()
    lvl0 = g[0][g[0]["octave"] == 0]
    want0 = ok[ok["octave"] == 0][: len(lvl0)]                  # the oracle's list is in raster order
    assert len(lvl0) == 434 + 256                               # level quota + slack
    assert lvl0.tobytes() == want0.tobytes()
    with pytest.raises(capi.VoError, match=stage):              # host-buffer entry point: error code
        ctx.orb_detect(a, 2000)
    ctx.sync()


def test_host_calls_report_only_their_own_truncation(ctx, oracle):
    """A device-resident call that cut a list and has not been synchronised yet must not make the next
    host-buffer call on clean images fail (that call answers for its own images only), and a
    host-buffer call on a tie-heavy image returns VO_ERR_CAPACITY together with the lists it kept."""
    import torch
    from visual_odometry_amd import capi
    ties = _dot_grid(12)
    a, b = synth.frame_pair(9)
    imgs = torch.from_numpy(np.stack([ties, a])).cuda()
    res = torch.zeros(capi.PAIR_DTYPE.itemsize, dtype=torch.uint8, device="cuda")
    ctx.pipeline_pairs_dev(imgs.data_ptr(), 1, 480, 640, 640, 480 * 640, 2000, 0.75, TUM_K, 1.0, 0.999, 0, res.data_ptr())
    k1, k2, qi, ti, d = ctx.match_images(a, b, 2000)              # clean pair: no error
    ok1, od1 = oracle.orb_detect(a, 2000)
    ok2, od2 = oracle.orb_detect(b, 2000)
    oq, ot, od = oracle.match(od1, od2)
    assert k1.tobytes() == ok1.tobytes() and k2.tobytes() == ok2.tobytes()
    assert np.array_equal(qi, oq) and np.array_equal(ti, ot) and np.array_equal(d, od)
    gk, gd = ctx.orb_detect(a, 2000)
    assert gk.tobytes() == ok1.tobytes() and np.array_equal(gd, od1)
    with pytest.raises(capi.VoError, match="Harris selection"):   # the earlier call's report is still due
        ctx.sync()
    with pytest.raises(capi.VoError, match="Harris selection"):
        ctx.match_images(ties, ties, 2000)
    k1, k2, qi, ti, d = ctx.match_images(ties, ties, 2000, allow_truncated=True)
    assert len(k1) == len(k2) > 434 + 256 and k1.tobytes() == k2.tobytes()   # (identical dots: no match passes the ratio test)
    gk, gd = ctx.orb_detect(ties, 2000, allow_truncated=True)
    assert gk.tobytes() == k1.tobytes() and len(gd) == len(gk)
    ctx.sync()                                                    # host-buffer calls leave nothing pending


# ------------------------------------------------------------------ RANSAC / pose
def _matched_points(oracle, seed, nfeat=2000):
    a, b = synth.frame_pair(seed)
    ka, da = oracle.orb_detect(a, nfeat)
    kb, db = oracle.orb_detect(b, nfeat)
    qi, ti, _ = oracle.match(da, db)
    return np.stack([ka["x"][qi], ka["y"][qi]], 1), np.stack([kb["x"][ti], kb["y"][ti]], 1)


@pytest.mark.parametrize("seed", [21, 22, 23])
def test_find_essential_shared_hypotheses(ctx, oracle, seed):
    p1, p2 = _matched_points(oracle, seed)
    og, oE, om, oit, osamp = oracle.find_essential(p1, p2, TUM_K)
    gg, gE, gm, git, gsamp = ctx.find_essential(p1, p2, TUM_K)
    assert git == oit
    assert np.array_equal(gsamp, osamp)                        # identical hypothesis index sets
    assert gg == og
    assert np.array_equal(gm, om)
    s = 1.0 if np.sum(gE * oE) > 0 else -1.0
    assert np.max(np.abs(gE - s * oE)) < 1e-8


def test_fixed_count_in_several_large_rounds(ctx):
    """A fixed hypothesis count is drawn up front and solved in one round; a batch whose pairs x count
    exceeds 2^18 hypotheses takes several such rounds.  Three pairs with 100 000 hypotheses each (two
    rounds) must give the records each pair gets alone (one round)."""
    import torch
    from visual_odometry_amd import capi
    frames = []
    for k in range(3):
        frames.extend(synth.frame_pair(700 + k))
    imgs = torch.from_numpy(np.stack(frames)).cuda()

    def run(first, n):
        res = torch.zeros(n * capi.PAIR_DTYPE.itemsize, dtype=torch.uint8, device="cuda")
        ctx.pipeline_pairs_dev(imgs[2 * first:].data_ptr(), n, 480, 640, 640, 480 * 640, 1000, 0.75, TUM_K, 1.0, 0.999,
                               100000, res.data_ptr())
        ctx.sync()
        return res.cpu().numpy().tobytes()

    together = run(0, 3)
    alone = b"".join(run(k, 1) for k in range(3))
    rec = np.frombuffer(together, capi.PAIR_DTYPE)
    assert np.all(rec["ransac_iters"] == 100000) and np.all(rec["valid"] == 1)
    assert together == alone


def test_root_kernel_variants_agree(ctx, options):
    """The root stage of the five-point solver exists as a one-wavefront kernel (the rounds' default)
    and as a 512-lane workgroup (second-chart pass, VO_OPT_SOLVE_VARIANT = 1): same pieces, same
    bookkeeping, so the RANSAC outcome must not depend on which one a round uses."""
    p1, p2 = None, None
    out = []
    for variant in (0, 1):
        options("solve_variant", variant)
        res = []
        for seed in (21, 24):
            a, b = synth.frame_pair(seed, sparse=True, moving_fraction=0.6, n_layers=24)
            k1, k2, qi, ti, _ = ctx.match_images(a, b, 2000)
            p1 = np.stack([k1["x"][qi], k1["y"][qi]], 1)
            p2 = np.stack([k2["x"][ti], k2["y"][ti]], 1)
            res.append(ctx.find_essential(p1, p2, TUM_K))
        out.append(res)
    for a, b in zip(*out):
        assert a[0] == b[0] and a[3] == b[3] and a[3] > 100          # inliers, iterations (several rounds)
        assert np.array_equal(a[2], b[2]) and np.array_equal(a[4], b[4])
        assert np.max(np.abs(a[1] - b[1])) < 1e-12


@pytest.mark.parametrize("fixed", [64, 700, 1500])
def test_find_essential_fixed_hypothesis_count(ctx, oracle, fixed):
    p1, p2 = _matched_points(oracle, 24, 1000)
    rng = np.random.default_rng(3)
    bad = rng.random(len(p1)) < 0.4                            # 40 % gross outliers
    p2 = p2.copy()
    p2[bad] += rng.uniform(-60, 60, (int(bad.sum()), 2)).astype(np.float32)
    og, oE, om, oit, osamp = oracle.find_essential(p1, p2, TUM_K, fixed_iters=fixed)
    gg, gE, gm, git, gsamp = ctx.find_essential(p1, p2, TUM_K, fixed_iters=fixed)
    assert git == oit == fixed
    assert np.array_equal(gsamp, osamp)
    assert gg == og and np.array_equal(gm, om)


@pytest.mark.parametrize("n", [6, 9, 15, 16, 17, 24, 40, 64, 200])
def test_sample_stream_with_few_points(ctx, oracle, n):
    # Few points make repeated indices inside a 5-subset frequent (the generator's rejection rule);
    # below 16 points one lane draws the plain way, from 16 on the stream is generated by 64 lanes
    # that jump ahead in the multiply-with-carry recurrence; at 16 and 17 points a round of 512 sets
    # needs more draws than are kept and ends the plain way.  700 fixed hypotheses = two rounds.
    p1, p2 = _matched_points(oracle, 27, 1000)
    sel = np.random.default_rng(n).choice(len(p1), n, replace=False)
    p1, p2 = p1[sel], p2[sel]
    og, oE, om, oit, osamp = oracle.find_essential(p1, p2, TUM_K, fixed_iters=700)
    gg, gE, gm, git, gsamp = ctx.find_essential(p1, p2, TUM_K, fixed_iters=700)
    assert git == oit == 700
    assert np.array_equal(gsamp, osamp)
    assert gg == og and np.array_equal(gm, om)


def _same_model_sets(ctx, oracle, p1, p2, K, samples):
    """Index of the first minimal sample on which the GPU's five-point stage and the oracle's report
    different model sets (E and -E are one model, 1e-7), or -1."""
    fx, fy, cx, cy = K
    n1 = np.stack([(p1[:, 0].astype(np.float64) - cx) / fx, (p1[:, 1].astype(np.float64) - cy) / fy], 1)
    n2 = np.stack([(p2[:, 0].astype(np.float64) - cx) / fx, (p2[:, 1].astype(np.float64) - cy) / fy], 1)
    x1, x2 = n1[samples], n2[samples]
    gpu = ctx.five_point(x1, x2)

    def one_sign(E):
        out = []
        for e in np.asarray(E).reshape(-1, 9):
            big = np.flatnonzero(np.abs(e) > 0.5 * np.max(np.abs(e)))[0]
            out.append(-e if e[big] < 0 else e)
        return np.array(out).reshape(-1, 9)

    for h in range(len(samples)):
        A, B = one_sign(gpu[h]), one_sign(oracle.five_point(x1[h], x2[h]))
        if len(A) != len(B) or (len(A) and max(np.min(np.max(np.abs(A - e), 1)) for e in B) > 1e-7):
            return h
    return -1


# The one scene of test_random_two_view_scenes on which GPU and oracle end differently, with what explains
# it (tools/enumerate_exceptions.py): case id -> ((GPU inliers, iterations), (oracle inliers, iterations),
# index of the first minimal sample whose model sets differ, that sample).
TWO_VIEW_EXCEPTIONS = {
    56: ((144, 600), (162, 600), 146, (296, 184, 63, 144, 237)),   # kind 0, 300 matches, 600 fixed hypotheses
}


def test_random_two_view_scenes(ctx, oracle):
    """Geometric scenes instead of matched ORB points: planar and small-rotation scenes, noise-free
    data on which every model of a hypothesis has all points as inliers (the winner among equals is
    the first in a chart-independent order), 5 to 3000 matches, up to 90 % outliers, fixed and
    adaptive hypothesis counts.  Sample sets, iteration count, inlier mask and model must be
    identical.  Both five-point stages approximate the complete solution set (the oracle by the union
    over four charts of the null space, the GPU by one chart and a second for odd model counts) and
    differ on about one minimal sample in a thousand (tools/dbg_model_sets.py); the scenes where such
    a sample changes the outcome are listed one by one in TWO_VIEW_EXCEPTIONS with the sample that
    explains them, and anything else fails.
    Inputs that are not well-posed (pure rotation, a baseline of a five-hundredth of the depth, sets
    full of repeated points) leave the minimal problem rank-deficient or nearly so and are skipped:
    which of a cluster of solutions a solver reports there is a matter of its root finder
    (tools/fuzz_ransac.py lists them)."""
    done, seen = 0, {}
    for case in range(110):
        rng = np.random.default_rng(1000003 * 7 + case)
        p1, p2, K, thr, fixed, (kind, noise, baseline) = synth.two_view_scene(rng)
        if kind in (3, 4) or baseline < 0.05:
            continue
        og, oE, om, oit, osamp = oracle.find_essential(p1, p2, K, threshold=thr, fixed_iters=fixed)
        gg, gE, gm, git, gsamp = ctx.find_essential(p1, p2, K, threshold=thr, fixed_iters=fixed)
        tag = f"case {case}: kind {kind} noise {noise} n {len(p1)} thr {thr} fixed {fixed}"
        k = min(git, oit)
        assert np.array_equal(gsamp[:k], osamp[:k]), tag          # one sample stream
        done += 1
        same = git == oit and gg == og and np.array_equal(gm, om)
        if same and og > 0:
            s = 1.0 if np.sum(gE * oE) > 0 else -1.0
            same = np.max(np.abs(gE - s * oE)) < 1e-8
        if not same:
            h = _same_model_sets(ctx, oracle, p1, p2, K, osamp[:k])
            seen[case] = ((gg, git), (og, oit), h, tuple(int(v) for v in osamp[h]) if h >= 0 else None)
            continue
        om_ = oracle.estimate_motion(p1, p2, K, threshold=thr, fixed_iters=fixed)
        gm_ = ctx.estimate_motion(p1, p2, K, threshold=thr, fixed_iters=fixed)
        assert bool(om_["valid"]) == bool(gm_["valid"]) and om_["inliers"] == gm_["inliers"], tag
        if om_["valid"]:
            assert np.max(np.abs(om_["R"] - gm_["R"])) < 1.7e-4 and np.max(np.abs(om_["t"] - gm_["t"])) < 1e-4, tag
    assert done >= 40
    assert seen == TWO_VIEW_EXCEPTIONS


@pytest.mark.parametrize("name", ["forward", "rotation", "forward_zyx", "rotation_zyx"])
def test_reference_motion_scenes(ctx, oracle, name):
    # ref:tests/test_motion_estimator.cpp:61-102 plus agreement with the oracle and the ground truth
    # (the _zyx files hold the points a g++ build of that test draws, the others a clang build's)
    z = np.load(os.path.join(GOLD, f"motion_scene_{name}.npz"))
    g = ctx.estimate_motion(z["p1"], z["p2"], KITTI_K)
    o = oracle.estimate_motion(z["p1"], z["p2"], KITTI_K)
    assert g["valid"] and g["inliers"] > 50
    if name.startswith("forward"):
        assert g["t"][2] > 0.9
    assert g["inliers"] == o["inliers"]
    assert _rot_err_deg(g["R"], o["R"]) < 0.01                 # degrees
    assert np.linalg.norm(g["t"] - o["t"]) < 1e-4              # unit-translation error
    assert _rot_err_deg(g["R"], z["R"]) < 0.01 and np.linalg.norm(g["t"] - z["t"]) < 1e-4


@pytest.mark.parametrize("seed", [21, 25, 26])
def test_estimate_motion_matches_oracle(ctx, oracle, seed):
    p1, p2 = _matched_points(oracle, seed)
    g = ctx.estimate_motion(p1, p2, TUM_K)
    o = oracle.estimate_motion(p1, p2, TUM_K)
    assert g["valid"] == o["valid"] and g["valid"]
    assert g["inliers"] == o["inliers"]
    assert np.array_equal(g["mask"], o["mask"])
    assert _rot_err_deg(g["R"], o["R"]) < 0.01
    assert np.linalg.norm(g["t"] - o["t"]) < 1e-4


def test_degenerate_pure_translation(ctx, oracle):
    # Sideways translation with integer disparities: y1 == y2 for every match, the
    # five-point polynomial has a double root at the true solution.  Both solvers must
    # still land on the same pose.
    a, b = synth.frame_pair(27, rot_deg=(0, 0, 0))
    ka, da = oracle.orb_detect(a, 2000)
    kb, db = oracle.orb_detect(b, 2000)
    qi, ti, _ = oracle.match(da, db)
    p1 = np.stack([ka["x"][qi], ka["y"][qi]], 1)
    p2 = np.stack([kb["x"][ti], kb["y"][ti]], 1)
    g = ctx.estimate_motion(p1, p2, TUM_K)
    o = oracle.estimate_motion(p1, p2, TUM_K)
    assert g["valid"] and o["valid"]
    assert g["inliers"] == o["inliers"] and np.array_equal(g["mask"], o["mask"])
    assert _rot_err_deg(g["R"], o["R"]) < 0.01 and np.linalg.norm(g["t"] - o["t"]) < 1e-4
    assert _rot_err_deg(g["R"], np.eye(3)) < 0.05 and abs(abs(g["t"][0]) - 1) < 1e-3


def test_five_point_models_match_oracle(ctx, oracle):
    rng = np.random.default_rng(77)
    n_sets, same = 256, 0
    x1 = rng.uniform(-0.6, 0.6, (n_sets, 5, 2))
    R = np.array([[0.9998, -0.0175, 0.01], [0.0176, 0.9998, -0.005], [-0.0099, 0.0052, 0.99994]])
    Z = rng.uniform(3, 12, (n_sets, 5, 1))
    X = np.concatenate([x1 * Z, Z], 2)
    Y = X @ R.T + np.array([0.3, -0.05, 0.1])
    x2 = Y[..., :2] / Y[..., 2:] + rng.normal(0, 1e-3, (n_sets, 5, 2))
    got = ctx.five_point(x1, x2)
    for i in range(n_sets):
        ref = oracle.five_point(x1[i], x2[i])
        for E in got[i]:                                   # every GPU model is a true solution
            r = [np.array([*x2[i, k], 1]) @ E @ np.array([*x1[i, k], 1]) for k in range(5)]
            assert np.max(np.abs(r)) < 1e-9 and abs(np.linalg.det(E)) < 1e-9
        if len(ref) == len(got[i]) and all(
                min(min(np.abs(E - F).max(), np.abs(E + F).max()) for F in ref) < 1e-8 for E in got[i]):
            same += 1
    assert same == n_sets   # (two independent root finders; the refinement and the second chart make the sets complete)


def test_five_point_with_a_repeated_point(ctx, oracle):
    # Two equal correspondences in a sample (keypoints of two pyramid levels at one position do
    # that in practice) leave four constraints: the degree-10 polynomial loses its leading
    # coefficient and the solutions form a family.  Whatever comes out must still be a solution,
    # and the call must not stall on such sets (they take the same parallel path as the others).
    rng = np.random.default_rng(78)
    n_sets, same = 128, 0
    x1 = rng.uniform(-0.6, 0.6, (n_sets, 5, 2))
    R = np.array([[0.9998, -0.0175, 0.01], [0.0176, 0.9998, -0.005], [-0.0099, 0.0052, 0.99994]])
    Z = rng.uniform(3, 12, (n_sets, 5, 1))
    X = np.concatenate([x1 * Z, Z], 2)
    Y = X @ R.T + np.array([0.3, -0.05, 0.1])
    x2 = Y[..., :2] / Y[..., 2:] + rng.normal(0, 1e-3, (n_sets, 5, 2))
    x1[:, 4], x2[:, 4] = x1[:, 1], x2[:, 1]
    got = ctx.five_point(x1, x2)
    n_models = 0
    for i in range(n_sets):
        ref = oracle.five_point(x1[i], x2[i])
        n_models += len(got[i])
        for E in got[i]:
            r = [np.array([*x2[i, k], 1]) @ E @ np.array([*x1[i, k], 1]) for k in range(5)]
            assert np.max(np.abs(r)) < 1e-9 and abs(np.linalg.det(E)) < 1e-9
        if len(ref) == len(got[i]) and all(
                min(min(np.abs(E - F).max(), np.abs(E + F).max()) for F in ref) < 1e-8 for E in got[i]):
            same += 1
    # (nothing is asserted about which members of the family come out: the oracle's four charts and
    # the GPU's two pick different ones - on no set of this test do the two lists coincide)
    assert n_models > 0


def test_motion_validity_rules(ctx):
    # ref:src/motion_estimator.cpp:52-54,83-85
    p1 = np.array([[100, 100], [200, 200], [300, 300]], np.float32)
    r = ctx.estimate_motion(p1, p1 + np.float32(10), KITTI_K)
    assert not r["valid"] and r["inliers"] == 0
    rng = np.random.default_rng(0)                              # pure noise: no consistent model
    a = rng.uniform(0, 600, (40, 2)).astype(np.float32)
    b = rng.uniform(0, 600, (40, 2)).astype(np.float32)
    r = ctx.estimate_motion(a, b, KITTI_K)
    assert r["inliers"] == 0 or r["valid"]


# ------------------------------------------------------------------ the large configurations
def test_full_hd_10k_keypoints_4096_hypotheses(ctx, oracle):
    # 1920x1080, 10 000 keypoints, fixed 4096-hypothesis RANSAC (no early exit)
    a, b = synth.frame_pair(61, height=1080, width=1920)
    K = synth.default_K(1920, 1080)
    ga, gda = ctx.orb_detect(a, 10000)
    oa, oda = oracle.orb_detect(a, 10000)
    assert len(ga) == len(oa) >= 9000
    assert ga.tobytes() == oa.tobytes() and np.array_equal(gda, oda)
    gb, gdb = ctx.orb_detect(b, 10000)
    gq, gt, gd = ctx.match_ratio(gda, gdb)
    oq, ot, od = oracle.match(gda, gdb)
    assert np.array_equal(gq, oq) and np.array_equal(gt, ot) and np.array_equal(gd, od)
    p1 = np.stack([ga["x"][gq], ga["y"][gq]], 1)
    p2 = np.stack([gb["x"][gt], gb["y"][gt]], 1)
    og, oE, om, oit, osamp = oracle.find_essential(p1, p2, K, fixed_iters=4096)
    gg, gE, gm, git, gsamp = ctx.find_essential(p1, p2, K, fixed_iters=4096)
    assert git == oit == 4096 and np.array_equal(gsamp, osamp)
    assert gg == og and np.array_equal(gm, om)
    g = ctx.estimate_motion(p1, p2, K, fixed_iters=4096)
    o = oracle.estimate_motion(p1, p2, K, fixed_iters=4096)
    assert g["valid"] and g["inliers"] == o["inliers"]
    assert _rot_err_deg(g["R"], o["R"]) < 0.01 and np.linalg.norm(g["t"] - o["t"]) < 1e-4


def test_kitti_shape_pipeline_pair(ctx, oracle):
    # 1241x376, 4000 keypoints through the fused device-resident path
    import torch
    from visual_odometry_amd import capi
    a, b = synth.frame_pair(62, height=376, width=1241)
    imgs = torch.from_numpy(np.stack([a, b])).cuda()
    res = torch.zeros(capi.PAIR_DTYPE.itemsize, dtype=torch.uint8, device="cuda")
    ctx.pipeline_pairs_dev(imgs.data_ptr(), 1, 376, 1241, 1241, 376 * 1241, 4000, 0.75, KITTI_K, 1.0, 0.999, 0,
                           res.data_ptr())
    ctx.sync()
    r = np.frombuffer(res.cpu().numpy().tobytes(), capi.PAIR_DTYPE)[0]
    ka, da = oracle.orb_detect(a, 4000)
    kb, db = oracle.orb_detect(b, 4000)
    qi, ti, _ = oracle.match(da, db)
    o = oracle.estimate_motion(np.stack([ka["x"][qi], ka["y"][qi]], 1), np.stack([kb["x"][ti], kb["y"][ti]], 1),
                               KITTI_K)
    assert (r["n_kp1"], r["n_kp2"], r["n_matches"]) == (len(ka), len(kb), len(qi))
    assert bool(r["valid"]) == o["valid"] and r["inliers"] == o["inliers"]
    assert _rot_err_deg(r["R"].reshape(3, 3), o["R"]) < 0.01 and np.linalg.norm(r["t"] - o["t"]) < 1e-4


def test_full_hd_fused_pipeline_fixed_4096_hypotheses(ctx, oracle):
    """The 1080p / 10 000-keypoint / fixed-4096-hypothesis configuration through the fused
    device-resident pipeline (two pairs), records against the oracle."""
    import torch
    from visual_odometry_amd import capi
    K = synth.default_K(1920, 1080)
    frames = list(synth.frame_pair(63, height=1080, width=1920)) + list(synth.frame_pair(64, height=1080, width=1920))
    imgs = torch.from_numpy(np.stack(frames)).cuda()
    res = torch.zeros(2 * capi.PAIR_DTYPE.itemsize, dtype=torch.uint8, device="cuda")
    ctx.pipeline_pairs_dev(imgs.data_ptr(), 2, 1080, 1920, 1920, 1080 * 1920, 10000, 0.75, K, 1.0, 0.999, 4096,
                           res.data_ptr())
    ctx.sync()
    out = np.frombuffer(res.cpu().numpy().tobytes(), capi.PAIR_DTYPE)
    for p in range(2):
        ka, da = oracle.orb_detect(frames[2 * p], 10000)
        kb, db = oracle.orb_detect(frames[2 * p + 1], 10000)
        qi, ti, _ = oracle.match(da, db)
        o = oracle.estimate_motion(np.stack([ka["x"][qi], ka["y"][qi]], 1), np.stack([kb["x"][ti], kb["y"][ti]], 1),
                                   K, fixed_iters=4096)
        r = out[p]
        assert (r["n_kp1"], r["n_kp2"], r["n_matches"]) == (len(ka), len(kb), len(qi))
        assert r["ransac_iters"] == 4096 and r["status"] == 0
        assert bool(r["valid"]) == o["valid"] and r["inliers"] == o["inliers"]
        assert _rot_err_deg(r["R"].reshape(3, 3), o["R"]) < 0.01 and np.linalg.norm(r["t"] - o["t"]) < 1e-4


def test_c1_sequence_of_101_frames_1000_keypoints(ctx, oracle):
    """The reference's own CPU-runnable configuration (640x480, orb_matcher{1000}, 100 consecutive
    pairs of a directory, TUM intrinsics; ref:src/main.cpp:144-184) on the synthetic 101-frame
    sequence the benchmark's c1 workload uses, through vo_pipeline_sequence_dev: every fifth
    pair's record against the oracle (which detects each frame on its own, as the reference does)."""
    import torch
    from visual_odometry_amd import capi
    frames = synth.sequence(0x5EED0000, 101)
    seq = torch.from_numpy(np.stack(frames)).cuda()
    res = torch.zeros(100 * capi.PAIR_DTYPE.itemsize, dtype=torch.uint8, device="cuda")
    ctx.pipeline_sequence_dev(seq.data_ptr(), 101, 480, 640, 640, 480 * 640, 1000, 0.75, TUM_K, 1.0, 0.999, 0,
                              res.data_ptr())
    ctx.sync()
    out = np.frombuffer(res.cpu().numpy().tobytes(), capi.PAIR_DTYPE)
    assert out["valid"].mean() > 0.9 and (out["status"] == 0).all()
    det = {}

    def detect(i):
        if i not in det:
            det[i] = oracle.orb_detect(frames[i], 1000)
        return det[i]

    for p in list(range(0, 100, 5)) + [99]:
        (ka, da), (kb, db) = detect(p), detect(p + 1)
        qi, ti, _ = oracle.match(da, db)
        o = oracle.estimate_motion(np.stack([ka["x"][qi], ka["y"][qi]], 1), np.stack([kb["x"][ti], kb["y"][ti]], 1), TUM_K)
        r = out[p]
        assert (r["n_kp1"], r["n_kp2"], r["n_matches"]) == (len(ka), len(kb), len(qi)), p
        assert bool(r["valid"]) == o["valid"] and r["inliers"] == o["inliers"], p
        if o["valid"]:
            assert _rot_err_deg(r["R"].reshape(3, 3), o["R"]) < 0.01 and np.linalg.norm(r["t"] - o["t"]) < 1e-4, p


@pytest.mark.parametrize("shape,nfeat", [((480, 640), 2000), ((376, 1241), 1000), ((301, 423), 300)])
def test_match_images_host_pair_call(ctx, oracle, shape, nfeat):
    """vo_match_images (one upload, one download, descriptors stay in HBM) against the oracle's
    detect x2 + match: keypoint records of both images byte for byte, matches in query order."""
    a, b = synth.frame_pair(90 + nfeat, height=shape[0], width=shape[1])
    k1, k2, qi, ti, d = ctx.match_images(a, b, nfeat, 0.75)
    oa, da = oracle.orb_detect(a, nfeat)
    ob, db = oracle.orb_detect(b, nfeat)
    oq, ot, od = oracle.match(da, db, 0.75)
    assert k1.tobytes() == oa.tobytes() and k2.tobytes() == ob.tobytes()
    assert len(oq) > 20
    assert np.array_equal(qi, oq) and np.array_equal(ti, ot) and np.array_equal(d, od)
    k1, k2, qi, ti, d = ctx.match_images(a, np.full(shape, 128, np.uint8), nfeat, 0.75)   # nothing to match against
    assert len(k1) == len(oa) and len(k2) == 0 and len(qi) == 0


# ------------------------------------------------------------------ fused pipeline
def test_pipeline_pairs_matches_oracle(ctx, oracle):
    import torch
    from visual_odometry_amd import capi
    n_pairs = 3
    frames = []
    for p in range(n_pairs):
        frames += list(synth.frame_pair(50 + p))
    imgs = torch.from_numpy(np.stack(frames)).cuda()
    res = torch.zeros(n_pairs * capi.PAIR_DTYPE.itemsize, dtype=torch.uint8, device="cuda")
    ctx.pipeline_pairs_dev(imgs.data_ptr(), n_pairs, 480, 640, 640, 480 * 640, 2000, 0.75, TUM_K, 1.0, 0.999, 0,
                           res.data_ptr())
    ctx.sync()
    torch.cuda.synchronize()
    out = np.frombuffer(res.cpu().numpy().tobytes(), capi.PAIR_DTYPE)
    for p in range(n_pairs):
        a, b = frames[2 * p], frames[2 * p + 1]
        ka, da = oracle.orb_detect(a, 2000)
        kb, db = oracle.orb_detect(b, 2000)
        qi, ti, _ = oracle.match(da, db)
        p1 = np.stack([ka["x"][qi], ka["y"][qi]], 1)
        p2 = np.stack([kb["x"][ti], kb["y"][ti]], 1)
        o = oracle.estimate_motion(p1, p2, TUM_K)
        r = out[p]
        assert (r["n_kp1"], r["n_kp2"], r["n_matches"]) == (len(ka), len(kb), len(qi))
        assert bool(r["valid"]) == o["valid"] and r["inliers"] == o["inliers"]
        assert _rot_err_deg(r["R"].reshape(3, 3), o["R"]) < 0.01
        assert np.linalg.norm(r["t"] - o["t"]) < 1e-4


@pytest.mark.gpu
@pytest.mark.parametrize("nfeat", [64, 128, 190])
def test_pipeline_tied_responses_return_more_than_max_features(ctx, oracle, nfeat):
    """A tiled texture repeats every corner, the detector keeps all ties of the last retained
    response (as the library's retainBest does), and an image ends up with more keypoints than
    max_features: every stage after the detector must carry the surplus (the capacity of the
    per-image tables is max_features plus slack, not max_features)."""
    import torch
    from visual_odometry_amd import capi
    rng = np.random.default_rng(8)
    blk = np.kron(rng.integers(0, 256, (24, 24), dtype=np.uint8), np.ones((5, 5), np.uint8))
    a = np.tile(blk, (5, 6))[:480, :640].copy()
    b = np.roll(a, (2, 3), (0, 1)).copy()
    ka, da = oracle.orb_detect(a, nfeat)
    kb, db = oracle.orb_detect(b, nfeat)
    assert len(ka) > nfeat, "the scene no longer produces a surplus: pick another texture"
    qi, ti, _ = oracle.match(da, db)
    imgs = torch.from_numpy(np.stack([a, b, b, a])).cuda()
    res = torch.zeros(2 * capi.PAIR_DTYPE.itemsize, dtype=torch.uint8, device="cuda")
    ctx.pipeline_pairs_dev(imgs.data_ptr(), 2, 480, 640, 640, 480 * 640, nfeat, 0.75, TUM_K, 1.0, 0.999, 0,
                           res.data_ptr())
    ctx.sync()
    out = np.frombuffer(res.cpu().numpy().tobytes(), capi.PAIR_DTYPE)
    assert (out[0]["n_kp1"], out[0]["n_kp2"], out[0]["n_matches"]) == (len(ka), len(kb), len(qi))
    qi2, _, _ = oracle.match(db, da)
    assert (out[1]["n_kp1"], out[1]["n_kp2"], out[1]["n_matches"]) == (len(kb), len(ka), len(qi2))
    o = oracle.estimate_motion(np.stack([ka["x"][qi], ka["y"][qi]], 1), np.stack([kb["x"][ti], kb["y"][ti]], 1), TUM_K)
    assert bool(out[0]["valid"]) == o["valid"] and out[0]["inliers"] == o["inliers"]


@pytest.mark.gpu
def test_pipeline_results_do_not_depend_on_matcher_variant(ctx):
    """The batched matcher launch (per-pair counts, strided descriptor tables) with every kernel
    variant forced: the pair records must be byte-identical to the default launch's, which the
    tests above check against the oracle."""
    import torch
    from visual_odometry_amd import capi
    frames = []
    for p in range(3):
        frames += list(synth.frame_pair(70 + p))
    imgs = torch.from_numpy(np.stack(frames)).cuda()

    def run():
        res = torch.zeros(3 * capi.PAIR_DTYPE.itemsize, dtype=torch.uint8, device="cuda")
        ctx.pipeline_pairs_dev(imgs.data_ptr(), 3, 480, 640, 640, 480 * 640, 1500, 0.75, TUM_K, 1.0, 0.999, 0,
                               res.data_ptr())
        ctx.sync()
        return res.cpu().numpy().tobytes()

    want = run()
    assert np.frombuffer(want, capi.PAIR_DTYPE)["n_matches"].min() > 50
    for knob in ["knn_waves=1", "knn_waves=4", "knn_waves=8", "knn_mode=0", "knn_mode=1", "knn_mode=3", "knn_queries_per_lane=2",
                 "knn_queries_per_lane=4"]:
        name, value = knob.split("=")
        before = ctx.get_option(name)
        ctx.set_option(name, int(value))
        assert run() == want, knob
        ctx.set_option(name, before)


@pytest.mark.gpu
def test_estimate_motion_repeated_calls_agree(ctx, oracle):
    """A call from host points whose predecessor ended within the first RANSAC round asks the GPU
    after its own first round whether it may stop launching: the same points must give the same
    answer whether the call came first (no such question), second or third, and whether the call
    before it was an easy or a hard one."""
    rng = np.random.default_rng(77)
    easy1 = rng.uniform(40, 600, (400, 2)).astype(np.float32)
    R = np.array([[0.9998, -0.01, 0.017], [0.01, 0.99995, 0.0], [-0.017, 0.0, 0.99985]])
    depth = rng.uniform(2, 6, 400)
    Kf = np.array([[TUM_K[0], 0, TUM_K[2]], [0, TUM_K[1], TUM_K[3]], [0, 0, 1]])
    X = (np.linalg.inv(Kf) @ np.c_[easy1, np.ones(400)].T) * depth
    x2 = Kf @ (R @ X + np.array([[0.1], [0.02], [0.01]]))
    easy2 = (x2[:2] / x2[2]).T.astype(np.float32)
    hard1 = easy1.copy()
    hard2 = easy2.copy()
    hard2[:300] = rng.uniform(40, 600, (300, 2)).astype(np.float32)   # a quarter inliers: hundreds of hypotheses

    def same(a, b):
        return a["valid"] == b["valid"] and a["inliers"] == b["inliers"] and np.array_equal(a["R"], b["R"]) and \
            np.array_equal(a["t"], b["t"]) and np.array_equal(a["mask"], b["mask"])

    ref_easy = ctx.estimate_motion(easy1, easy2, TUM_K)
    ref_hard = ctx.estimate_motion(hard1, hard2, TUM_K)
    assert ref_easy["valid"] and ref_easy["inliers"] > 300
    want = oracle.estimate_motion(easy1, easy2, TUM_K)
    assert want["inliers"] == ref_easy["inliers"]
    for seq in ("eee", "ehe", "hee", "ehhe"):
        for kind in seq:
            got = ctx.estimate_motion(easy1, easy2, TUM_K) if kind == "e" else ctx.estimate_motion(hard1, hard2, TUM_K)
            assert same(got, ref_easy if kind == "e" else ref_hard), (seq, kind)


@pytest.mark.gpu
def test_pipeline_large_calls_equal_small_ones(ctx, options):
    """Between 64 and 256 pairs of 2000 keypoints a pipeline call enqueues its kNN search on the tail
    stream, beside the next call's detection, instead of on the main stream.  Three such calls back to back
    (70, 66 and 70 pairs, two record buffers) must give the records the same pairs give in calls of
    seven, which keep the search on the main stream - also for the sequence form."""
    import torch
    from visual_odometry_amd import capi
    frames = []
    for p in range(70):
        frames += list(synth.frame_pair(7000 + p) if p % 4 else
                       synth.frame_pair(0x5EED0000 + 2300 + p, sparse=True, moving_fraction=0.6, n_layers=24))
    imgs = torch.from_numpy(np.stack(frames)).cuda()
    rec = capi.PAIR_DTYPE.itemsize
    want = torch.zeros(70 * rec, dtype=torch.uint8, device="cuda")
    for p0 in range(0, 70, 7):
        ctx.pipeline_pairs_dev(imgs.data_ptr() + 2 * p0 * 480 * 640, 7, 480, 640, 640, 480 * 640, 2000, 0.75, TUM_K, 1.0,
                               0.999, 0, want.data_ptr() + p0 * rec)
    ctx.sync()
    want = np.frombuffer(want.cpu().numpy().tobytes(), capi.PAIR_DTYPE)
    assert (want["valid"] == 1).all() and want["ransac_iters"].max() > 100 and want["ransac_iters"].min() < 40
    bufs = [torch.zeros(70 * rec, dtype=torch.uint8, device="cuda") for _ in range(2)]
    plan = [(0, 70, 0), (4, 66, 1), (0, 70, 0)]   # (first pair, pairs, record buffer): the third overwrites the first's
    for p0, n, b in plan:
        ctx.pipeline_pairs_dev(imgs.data_ptr() + 2 * p0 * 480 * 640, n, 480, 640, 640, 480 * 640, 2000, 0.75, TUM_K, 1.0,
                               0.999, 0, bufs[b].data_ptr())
    ctx.sync()
    got0 = np.frombuffer(bufs[0].cpu().numpy().tobytes(), capi.PAIR_DTYPE)
    got1 = np.frombuffer(bufs[1].cpu().numpy().tobytes(), capi.PAIR_DTYPE)[:66]
    assert got0.tobytes() == want.tobytes()
    assert got1.tobytes() == want[4:].tobytes()
    # the opt-in graph replay of such a call (captured on its second occurrence, replayed on the third)
    options("pipeline_graph", 1)
    for rep in range(3):
        bufs[0].zero_()
        ctx.pipeline_pairs_dev(imgs.data_ptr(), 70, 480, 640, 640, 480 * 640, 2000, 0.75, TUM_K, 1.0, 0.999, 0,
                               bufs[0].data_ptr())
        ctx.sync()
        assert bufs[0].cpu().numpy().tobytes() == want.tobytes(), rep
    options("pipeline_graph", 0)
    # the sequence form: 140 frames -> 139 pairs in one call, against calls of at most 20 frames
    seq_want = torch.zeros(139 * rec, dtype=torch.uint8, device="cuda")
    for f0 in range(0, 139, 19):
        nf = min(20, 140 - f0)
        ctx.pipeline_sequence_dev(imgs.data_ptr() + f0 * 480 * 640, nf, 480, 640, 640, 480 * 640, 2000, 0.75, TUM_K, 1.0,
                                  0.999, 0, seq_want.data_ptr() + f0 * rec)
    seq_got = torch.zeros(139 * rec, dtype=torch.uint8, device="cuda")
    ctx.pipeline_sequence_dev(imgs.data_ptr(), 140, 480, 640, 640, 480 * 640, 2000, 0.75, TUM_K, 1.0, 0.999, 0,
                              seq_got.data_ptr())
    ctx.sync()
    assert seq_got.cpu().numpy().tobytes() == seq_want.cpu().numpy().tobytes()


@pytest.mark.gpu
def test_pipeline_graph_replay_equals_launches(ctx, options):
    """VO_OPT_PIPELINE_GRAPH: a pipeline call that repeats the previous one's arguments is captured
    as a HIP graph on its second occurrence and replayed from the third on.  Replays must give
    the records of the ordinary launch chain, also after a different call in between (which
    retires the graph), around an estimate_motion call on the shared workspace, with tickets that
    can be waited for, and for the sequence form."""
    import torch
    from visual_odometry_amd import capi
    frames_a, frames_b = [], []
    for p in range(3):
        frames_a += list(synth.frame_pair(900 + p))
    for p in range(2):
        frames_b += list(synth.frame_pair(950 + p))
    imgs_a = torch.from_numpy(np.stack(frames_a)).cuda()
    imgs_b = torch.from_numpy(np.stack(frames_b)).cuda()
    res_a = torch.zeros(3 * capi.PAIR_DTYPE.itemsize, dtype=torch.uint8, device="cuda")
    res_b = torch.zeros(2 * capi.PAIR_DTYPE.itemsize, dtype=torch.uint8, device="cuda")
    res_s = torch.zeros(5 * capi.PAIR_DTYPE.itemsize, dtype=torch.uint8, device="cuda")

    def run_a():
        ctx.pipeline_pairs_dev(imgs_a.data_ptr(), 3, 480, 640, 640, 480 * 640, 1500, 0.75, TUM_K, 1.0, 0.999, 0,
                               res_a.data_ptr())

    def run_b():
        ctx.pipeline_pairs_dev(imgs_b.data_ptr(), 2, 480, 640, 640, 480 * 640, 1500, 0.75, TUM_K, 1.0, 0.999, 0,
                               res_b.data_ptr())

    def run_s():   # six frames as a sequence: five pairs
        ctx.pipeline_sequence_dev(imgs_a.data_ptr(), 6, 480, 640, 640, 480 * 640, 1500, 0.75, TUM_K, 1.0, 0.999, 0,
                                  res_s.data_ptr())

    def fetch(t):
        ctx.sync()
        return t.cpu().numpy().tobytes()

    run_a(); want_a = fetch(res_a)
    run_b(); want_b = fetch(res_b)
    run_s(); want_s = fetch(res_s)
    assert (np.frombuffer(want_a, capi.PAIR_DTYPE)["inliers"] > 50).all()
    rng = np.random.default_rng(5)
    p1 = rng.uniform(50, 590, (300, 2)).astype(np.float32)
    p2 = p1 + rng.normal(0, 3, p1.shape).astype(np.float32)
    ref_motion = ctx.estimate_motion(p1, p2, TUM_K)

    options("pipeline_graph", 1)
    for rep in range(5):          # ordinary, captured + replayed, replayed ...
        res_a.zero_()
        run_a()
        assert fetch(res_a) == want_a, rep
    res_b.zero_()
    run_b()                       # another call: the graph is retired, this one runs the ordinary way
    assert fetch(res_b) == want_b
    for rep in range(3):          # a burst without a fence in between, then results by ticket
        res_a.zero_()
        torch.cuda.synchronize()
        run_a()
        run_a()
        run_a()
        ticket = ctx.pipeline_ticket()
        ctx.pipeline_wait(ticket, results=True)
        assert res_a.cpu().numpy().tobytes() == want_a
        mid = ctx.estimate_motion(p1, p2, TUM_K)
        assert mid["inliers"] == ref_motion["inliers"] and np.array_equal(mid["R"], ref_motion["R"])
    for rep in range(4):
        res_s.zero_()
        run_s()
        assert fetch(res_s) == want_s, rep
    options("pipeline_graph", 0)
    res_a.zero_()
    run_a()
    assert fetch(res_a) == want_a


@pytest.mark.gpu
def test_pipeline_back_to_back_calls_are_independent(ctx):
    """Calls only enqueue (the RANSAC rounds of one call overlap the detection of the next on a
    second stream): a burst of calls with different batches, one sync at the end, must give each
    batch the results it gets when run alone - also when an estimate_motion call (which shares
    the RANSAC workspace) lands in the middle of the burst."""
    import torch
    from visual_odometry_amd import capi
    batches = []
    for s in range(4):
        n_pairs = 2 + (s % 2)
        frames = []
        for p in range(n_pairs):
            frames += list(synth.frame_pair(300 + 10 * s + p))
        batches.append((n_pairs, torch.from_numpy(np.stack(frames)).cuda()))

    def run(n_pairs, imgs, res):
        ctx.pipeline_pairs_dev(imgs.data_ptr(), n_pairs, 480, 640, 640, 480 * 640, 2000, 0.75, TUM_K, 1.0, 0.999, 0,
                               res.data_ptr())

    alone = []
    for n_pairs, imgs in batches:
        res = torch.zeros(n_pairs * capi.PAIR_DTYPE.itemsize, dtype=torch.uint8, device="cuda")
        run(n_pairs, imgs, res)
        ctx.sync()
        alone.append(np.frombuffer(res.cpu().numpy().tobytes(), capi.PAIR_DTYPE).copy())
        assert alone[-1]["valid"].all() and (alone[-1]["inliers"] > 50).all()

    rng = np.random.default_rng(5)
    p1 = rng.uniform(50, 590, (300, 2)).astype(np.float32)
    p2 = p1 + rng.normal(0, 3, p1.shape).astype(np.float32)
    ref_motion = ctx.estimate_motion(p1, p2, TUM_K)

    outs = [torch.zeros(n * capi.PAIR_DTYPE.itemsize, dtype=torch.uint8, device="cuda") for n, _ in batches]
    for rep in range(3):
        for o in outs:
            o.zero_()
        torch.cuda.synchronize()
        run(*batches[0], outs[0])
        run(*batches[1], outs[1])
        mid = ctx.estimate_motion(p1, p2, TUM_K)   # host-blocking call on the shared workspace
        run(*batches[2], outs[2])
        run(*batches[3], outs[3])
        ctx.sync()
        for o, a in zip(outs, alone):
            got = np.frombuffer(o.cpu().numpy().tobytes(), capi.PAIR_DTYPE)
            assert got.tobytes() == a.tobytes()
        for k in ("valid", "inliers"):
            assert mid[k] == ref_motion[k]
        assert np.array_equal(mid["R"], ref_motion["R"]) and np.array_equal(mid["t"], ref_motion["t"])


@pytest.mark.gpu
def test_pipeline_calls_sharing_result_records_stay_ordered(ctx):
    """Consecutive calls run their tails on alternate streams; when they are handed the same
    result records the later call's records must be what is left - also when the earlier call
    is the larger, slower one - while calls with records of their own each keep theirs."""
    import torch
    from visual_odometry_amd import capi

    def batch(seeds, sparse):
        frames = []
        for s_ in seeds:
            frames += list(synth.frame_pair(s_, sparse=True, moving_fraction=0.6, n_layers=24) if sparse
                           else synth.frame_pair(s_))
        return len(seeds), torch.from_numpy(np.stack(frames)).cuda()

    def run(b, res):
        ctx.pipeline_pairs_dev(b[1].data_ptr(), b[0], 480, 640, 640, 480 * 640, 2000, 0.75, TUM_K, 1.0, 0.999, 0,
                               res.data_ptr())

    def records(res, n):
        return np.frombuffer(res.cpu().numpy().tobytes(), capi.PAIR_DTYPE)[:n].copy()

    big, small = batch(range(700, 706), True), batch(range(710, 712), False)   # many RANSAC rounds / few
    alone = []
    for b_ in (big, small):
        res = torch.zeros(6 * capi.PAIR_DTYPE.itemsize, dtype=torch.uint8, device="cuda")
        run(b_, res)
        ctx.sync()
        alone.append(records(res, b_[0]))
    assert alone[0]["ransac_iters"].mean() > 3 * alone[1]["ransac_iters"].mean()
    shared = torch.zeros(6 * capi.PAIR_DTYPE.itemsize, dtype=torch.uint8, device="cuda")
    own = torch.zeros(6 * capi.PAIR_DTYPE.itemsize, dtype=torch.uint8, device="cuda")
    for rep in range(3):
        shared.zero_()
        own.zero_()
        torch.cuda.synchronize()
        run(big, shared)
        run(small, shared)      # same records: ordered behind the first call's tail
        run(big, own)           # records of its own: free to run beside the second call's tail
        ctx.sync()
        got = records(shared, 6)
        assert got[:2].tobytes() == alone[1].tobytes()
        assert got[2:].tobytes() == alone[0][2:].tobytes()
        assert records(own, 6).tobytes() == alone[0].tobytes()


@pytest.mark.gpu
def test_pipeline_tickets_release_inputs_and_results_per_call(ctx):
    """vo_pipeline_wait: after VO_WAIT_INPUTS for a call its image buffer may be overwritten while
    the call's RANSAC rounds (and a younger call) are still running; after VO_WAIT_RESULTS its
    records are final although a younger call is still in flight.  Tickets count up per context,
    unknown ones are refused, old ones are waited for through a younger call."""
    import torch
    from visual_odometry_amd import capi
    sets = []
    for s_ in range(6):
        frames = []
        for p in range(3):
            frames += list(synth.frame_pair(400 + 10 * s_ + p))
        sets.append(np.stack(frames))

    def launch(imgs, res):
        ctx.pipeline_pairs_dev(imgs.data_ptr(), 3, 480, 640, 640, 480 * 640, 2000, 0.75, TUM_K, 1.0, 0.999, 0, res.data_ptr())

    def records(res):
        return np.frombuffer(res.cpu().numpy().tobytes(), capi.PAIR_DTYPE).copy()

    alone = []
    for st in sets:
        imgs = torch.from_numpy(st).cuda()
        res = torch.zeros(3 * capi.PAIR_DTYPE.itemsize, dtype=torch.uint8, device="cuda")
        launch(imgs, res)
        ctx.sync()
        alone.append(records(res))
    t0 = ctx.pipeline_ticket()
    assert t0 >= len(sets)

    bufs = [torch.from_numpy(sets[0]).cuda(), torch.from_numpy(sets[1]).cuda()]
    outs = [torch.zeros(3 * capi.PAIR_DTYPE.itemsize, dtype=torch.uint8, device="cuda") for _ in sets]
    tickets = []
    for k in range(len(sets)):
        launch(bufs[k % 2], outs[k])
        tickets.append(ctx.pipeline_ticket())
        assert tickets[-1] == t0 + k + 1
        if k >= 1:
            # the buffer of call k - 1 takes the images of call k + 1 as soon as call k - 1 has read it
            ctx.pipeline_wait(tickets[k - 1], inputs=True)
            if k + 1 < len(sets):
                bufs[(k + 1) % 2].copy_(torch.from_numpy(sets[k + 1]).cuda())
            else:
                bufs[(k + 1) % 2].fill_(7)
            torch.cuda.synchronize()   # (the copy ran on torch's stream)
            ctx.pipeline_wait(tickets[k - 1], results=True)
            got = records(outs[k - 1])
            assert got.tobytes() == alone[k - 1].tobytes(), f"call {k - 1}"
    ctx.pipeline_wait(tickets[0], inputs=True, results=True)   # five calls old: served by a younger call's marks
    ctx.pipeline_wait(tickets[-1], results=True)
    assert records(outs[-1]).tobytes() == alone[-1].tobytes()
    ctx.sync()
    for bad in (0, tickets[-1] + 1):
        with pytest.raises(RuntimeError, match="vo_pipeline_wait"):
            ctx.pipeline_wait(bad, results=True)
    with pytest.raises(RuntimeError, match="vo_pipeline_wait"):
        ctx.pipeline_wait(tickets[-1])


@pytest.mark.gpu
def test_pipeline_sequence_equals_pairwise(ctx):
    """A sequence of frames gives, for each consecutive pair, exactly the record the pair
    pipeline gives for that pair (each frame is detected once instead of twice)."""
    import torch
    from visual_odometry_amd import capi
    frames = [synth.frame_pair(400)[0]]
    for s in range(5):
        frames.append(synth.frame_pair(400 + s)[1] if s % 2 == 0 else synth.frame_pair(401 + s)[0])
    n = len(frames)
    seq = torch.from_numpy(np.stack(frames)).cuda()
    pairs = torch.from_numpy(np.stack([frames[i + k] for i in range(n - 1) for k in range(2)])).cuda()
    r_seq = torch.zeros((n - 1) * capi.PAIR_DTYPE.itemsize, dtype=torch.uint8, device="cuda")
    r_pair = torch.zeros_like(r_seq)
    ctx.pipeline_sequence_dev(seq.data_ptr(), n, 480, 640, 640, 480 * 640, 1500, 0.75, TUM_K, 1.0, 0.999, 0,
                              r_seq.data_ptr())
    ctx.sync()
    ctx.pipeline_pairs_dev(pairs.data_ptr(), n - 1, 480, 640, 640, 480 * 640, 1500, 0.75, TUM_K, 1.0, 0.999, 0,
                           r_pair.data_ptr())
    ctx.sync()
    a = np.frombuffer(r_seq.cpu().numpy().tobytes(), capi.PAIR_DTYPE)
    b = np.frombuffer(r_pair.cpu().numpy().tobytes(), capi.PAIR_DTYPE)
    assert (a["n_kp1"] > 500).all() and (a["n_matches"] >= 0).all()
    assert a.tobytes() == b.tobytes()


@pytest.mark.gpu
def test_host_calls_between_async_pipeline_calls(ctx, oracle):
    """The pipeline's tail (ratio test, RANSAC) runs on a side stream and still reads the kNN table
    and the keypoint records of its batch; host-buffer calls that reuse those buffers on the
    main stream (matcher, detector) must wait for it, and must not disturb the batches in flight."""
    import torch
    from visual_odometry_amd import capi
    frames = []
    for p in range(3):
        frames += list(synth.frame_pair(520 + p))
    imgs = torch.from_numpy(np.stack(frames)).cuda()

    def run(res):
        ctx.pipeline_pairs_dev(imgs.data_ptr(), 3, 480, 640, 640, 480 * 640, 2000, 0.75, TUM_K, 1.0, 0.999, 0,
                               res.data_ptr())

    ref = torch.zeros(3 * capi.PAIR_DTYPE.itemsize, dtype=torch.uint8, device="cuda")
    run(ref)
    ctx.sync()
    ref = ref.cpu().numpy().tobytes()
    q, t = synth.descriptors(31, 700), synth.descriptors(32, 900)
    oi, od = oracle.knn2(q, t)
    img = synth.frame_pair(77)[0]
    ok, odesc = oracle.orb_detect(img, 800)
    for rep in range(3):
        r1 = torch.zeros(3 * capi.PAIR_DTYPE.itemsize, dtype=torch.uint8, device="cuda")
        r2 = torch.zeros_like(r1)
        torch.cuda.synchronize()
        run(r1)
        gi, gd = ctx.hamming_knn2(q, t)          # shares the kNN table with the batch in flight
        run(r2)
        gk, gdesc = ctx.orb_detect(img, 800)     # re-plans the detector for another batch shape
        ctx.sync()
        assert np.array_equal(gi, oi) and np.array_equal(gd, od)
        assert gk.tobytes() == ok.tobytes() and np.array_equal(gdesc, odesc)
        assert r1.cpu().numpy().tobytes() == ref
        assert r2.cpu().numpy().tobytes() == ref


@pytest.mark.gpu
@pytest.mark.parametrize("n_ctx,n_frames", [(2, 9), (3, 8), (4, 3)])
def test_sequence_over_several_contexts_equals_one(ctx, n_ctx, n_frames):
    """vo_pipeline_sequence_multi: contiguous blocks of pairs on one context and host thread each,
    the boundary frame detected on both sides.  The box has one GPU, so the contexts all sit on
    device 0 - the split, the overlap, the threads and the ordering are what is exercised - and
    the records must be those of a single call, byte for byte (also with more contexts than pairs)."""
    from visual_odometry_amd import capi
    frames = []
    for s in range(n_frames):
        frames.append(synth.frame_pair(800 + s // 2)[s % 2])
    frames = np.stack(frames)
    want = ctx.pipeline_sequence(frames, 1500, TUM_K)
    ctxs = [capi.Context(0) for _ in range(n_ctx)]
    got = capi.pipeline_sequence_multi(ctxs, frames, 1500, TUM_K)
    assert got.tobytes() == want.tobytes()
    with pytest.raises(capi.VoError, match="listed twice"):
        capi.pipeline_sequence_multi([ctxs[0], ctxs[0]], frames, 1500, TUM_K)
    for c in ctxs:
        c.close()
