"""CPU suite: the oracle against the reference's own test contract, against
independent numpy restatements and against hand-derived vectors.  No GPU."""
import os

import numpy as np
import pytest

from visual_odometry_amd import synth

GOLD = os.path.join(os.path.dirname(__file__), "golden")
KITTI_K = (718.856, 718.856, 607.1928, 185.2157)


# ---- detector: ref:tests/test_feature_detector.cpp ---------------------------------------
def test_detects_keypoints_and_descriptors(oracle):
    img = synth.checkerboard(seed=1)
    kps, desc = oracle.orb_detect(img, 2000)
    assert len(kps) > 0
    assert desc.shape == (len(kps), 32)                # rows == keypoints, cols == 32
    k2, none = oracle.orb_detect(img, 2000, want_desc=False)
    assert none is None and np.array_equal(k2, kps)     # detect_keypoints_only == same keypoints


def test_respects_max_features(oracle):
    kps, _ = oracle.orb_detect(synth.checkerboard(seed=1), 100, want_desc=False)
    assert 0 < len(kps) <= 100


def test_empty_and_uniform_image(oracle):
    kps, _ = oracle.orb_detect(np.zeros((0, 0), np.uint8))
    assert len(kps) == 0
    kps, _ = oracle.orb_detect(np.full((480, 640), 128, np.uint8))
    assert len(kps) < 10


def test_pyramid_geometry_and_quota(oracle):
    sizes = [oracle.level_geometry(480, 640, l)[:2] for l in range(8)]
    assert sizes == [(640, 480), (533, 400), (444, 333), (370, 278), (309, 231), (257, 193), (214, 161), (179, 134)]
    q = oracle.features_per_level(2000)
    assert list(q) == [434, 362, 302, 251, 209, 175, 145, 122] and q.sum() == 2000


def test_keypoint_fields(oracle):
    a, _ = synth.frame_pair(3)
    kps, desc = oracle.orb_detect(a, 1000)
    assert np.all(kps["class_id"] == -1)
    assert np.all((kps["angle"] >= 0) & (kps["angle"] < 360))
    scale = np.float32(1.2) ** kps["octave"].astype(np.float64)
    assert np.allclose(kps["size"], 31 * scale, rtol=1e-5)
    # every keypoint lies >= 31 level-pixels from its level's border
    for l in range(8):
        w, h, s = oracle.level_geometry(480, 640, l)
        m = kps["octave"] == l
        x, y = kps["x"][m] / s, kps["y"][m] / s
        assert np.all((x > 30.5) & (x < w - 31 + 0.5) & (y > 30.5) & (y < h - 31 + 0.5))
    # canonical order: level-major, raster inside a level
    key = kps["octave"].astype(np.int64) << 40 | np.rint(kps["y"] / scale).astype(np.int64) << 20 | np.rint(
        kps["x"] / scale).astype(np.int64)
    assert np.all(np.diff(key) > 0)


def test_libstdcxx_order_is_a_permutation(oracle):
    a, _ = synth.frame_pair(4)
    k0, d0 = oracle.orb_detect(a, 500, order_mode=0)
    k1, d1 = oracle.orb_detect(a, 500, order_mode=1)
    assert len(k0) == len(k1)
    o0 = np.lexsort((k0["x"], k0["y"], k0["octave"]))
    o1 = np.lexsort((k1["x"], k1["y"], k1["octave"]))
    assert np.array_equal(k0[o0], k1[o1]) and np.array_equal(d0[o0], d1[o1])


# ---- hand-derived micro-vectors ---------------------------------------------------------------
def _patch(center, ring):
    """16x16 patch, value `center` everywhere, ring pixels of (8,8) set from `ring`."""
    dx = [0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1]
    dy = [3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1, 0, 1, 2, 3]
    p = np.full((16, 16), center, np.uint8)
    for k in range(16):
        p[8 + dy[k], 8 + dx[k]] = ring[k]
    return p


def test_fast_arc_length_and_threshold(oracle):
    c = 100
    ring9 = [c + 30] * 9 + [c] * 7          # exactly 9 contiguous brighter pixels
    ring8 = [c + 30] * 8 + [c] * 8          # 8: not a corner
    got9 = oracle.fast(_patch(c, ring9))
    assert any((r[0], r[1]) == (8, 8) for r in got9)
    s = [r[2] for r in got9 if (r[0], r[1]) == (8, 8)][0]
    assert s == 29                                      # largest t with |d| > t on the arc is 29
    assert not any((r[0], r[1]) == (8, 8) for r in oracle.fast(_patch(c, ring8)))
    # contrast of exactly 20 is not beyond the threshold, 21 is
    assert not any((r[0], r[1]) == (8, 8) for r in oracle.fast(_patch(c, [c + 20] * 9 + [c] * 7)))
    assert any((r[0], r[1]) == (8, 8) for r in oracle.fast(_patch(c, [c + 21] * 9 + [c] * 7)))
    # dark arc, wrapping around index 15 -> 0
    ring = [c] * 16
    for k in (13, 14, 15, 0, 1, 2, 3, 4, 5):
        ring[k] = c - 40
    got = oracle.fast(_patch(c, ring))
    assert [r[2] for r in got if (r[0], r[1]) == (8, 8)] == [39]


def test_fast_nms_equal_neighbours_both_dropped(oracle):
    # two horizontally adjacent corners with identical scores: strict '>' drops both
    img = np.full((24, 24), 100, np.uint8)
    img[:, 12:] = 100
    a = np.full((24, 24), 50, np.uint8)
    a[8:16, 8:16] = 200
    res = oracle.fast(a)
    sc = {(r[0], r[1]): r[2] for r in res}
    for (x, y), s in sc.items():
        for ddx in (-1, 0, 1):
            for ddy in (-1, 0, 1):
                if (ddx or ddy) and (x + ddx, y + ddy) in sc:
                    pytest.fail("two adjacent keypoints survived non-max suppression")


def test_fast_atan2_polynomial(oracle):
    # float32 evaluation of the published polynomial, independent of the C code
    def ref(y, x):
        f = np.float32
        k = f(np.float32(180) / f(np.pi)) if False else f(180 / np.pi)
        p1, p3, p5, p7 = (f(0.9997878412794807) * k, f(-0.3258083974640975) * k, f(0.1555786518463281) * k,
                          f(-0.04432655554792128) * k)
        ax, ay = abs(f(x)), abs(f(y))
        eps = f(2.220446049250313e-16)
        if ax >= ay:
            c = ay / (ax + eps)
            c2 = c * c
            a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c
        else:
            c = ax / (ay + eps)
            c2 = c * c
            a = f(90.0) - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c
        if x < 0:
            a = f(180.0) - a
        if y < 0:
            a = f(360.0) - a
        return a

    rng = np.random.default_rng(0)
    for _ in range(2000):
        y, x = rng.integers(-200000, 200000, 2)
        got = oracle.fast_atan2(float(y), float(x))
        assert np.float32(got) == ref(float(y), float(x))
        true = np.degrees(np.arctan2(float(y), float(x))) % 360.0
        assert abs(((got - true) + 180) % 360 - 180) < 0.3


def test_resize_blur_closed_forms(oracle):
    # a constant image stays constant through the exact resize; the 7x7 blur's taps sum to
    # 257/256 per axis, so a constant c maps to round_half_even(c * 257^2 / 65536) (saturated)
    for c in (0, 1, 77, 200, 255):
        img = np.full((120, 160), c, np.uint8)
        for l in (1, 3):
            assert np.all(oracle.pyramid_level(img, l) == c)
        exp = min(255, int(np.rint(c * 257 * 257 / 65536.0)))
        assert np.all(oracle.pyramid_level(img, 0, blurred=True) == exp)
    # horizontal ramp: bilinear resampling of a linear function stays within 1 of linear
    ramp = np.tile(np.arange(160, dtype=np.uint8), (120, 1))
    lv = oracle.pyramid_level(ramp, 1).astype(np.float64)
    w = lv.shape[1]
    ideal = (np.arange(w) + 0.5) * (160 / w) - 0.5
    assert np.max(np.abs(lv[5, 1:-1] - ideal[1:-1])) <= 1.0


# ---- matcher: ref:tests/test_feature_matcher.cpp --------------------------------------------
def _numpy_knn2(q, t):
    d = np.unpackbits(q[:, None, :] ^ t[None, :, :], axis=2).sum(2).astype(np.int32)
    order = np.argsort(d, axis=1, kind="stable")[:, :2]      # stable: ties -> lower train index
    idx = np.full((len(q), 2), -1, np.int32)
    dist = np.full((len(q), 2), -1, np.int32)
    k = order.shape[1]
    idx[:, :k] = order
    dist[:, :k] = np.take_along_axis(d, order, 1)
    return idx, dist


def test_knn2_against_numpy_with_ties(oracle):
    q = synth.descriptors(10, 300, dup_fraction=0.2)
    t = np.concatenate([synth.descriptors(11, 400, dup_fraction=0.3), q[:50], q[:50]])  # exact duplicates
    gi, gd = oracle.knn2(q, t)
    ri, rd = _numpy_knn2(q, t)
    assert np.array_equal(gi, ri) and np.array_equal(gd, rd)


def test_knn2_single_train_row_and_ratio_guard(oracle):
    q = synth.descriptors(12, 10)
    t = synth.descriptors(13, 1)
    gi, gd = oracle.knn2(q, t)
    assert np.all(gi[:, 0] == 0) and np.all(gi[:, 1] == -1)
    assert len(oracle.match(q, t)[0]) == 0       # fewer than two neighbours -> no match survives
    assert len(oracle.match(q[:0], t)[0]) == 0 and len(oracle.match(q, t[:0])[0]) == 0


def test_match_consistency_and_ratio_monotone(oracle):
    a = synth.checkerboard(seed=2, noise_sigma=15.0)
    b = synth.checkerboard(seed=3, noise_sigma=15.0, shift=10)
    ka, da = oracle.orb_detect(a)
    kb, db = oracle.orb_detect(b)
    qi, ti, d = oracle.match(da, db, 0.75)
    assert len(qi) > 0 and np.all(np.diff(qi) > 0)          # query order
    idx, dist = oracle.knn2(da, db)
    assert np.array_equal(ti, idx[qi, 0]) and np.array_equal(d, dist[qi, 0].astype(np.float32))
    loose = len(oracle.match(da, db, 0.9)[0])
    strict = len(oracle.match(da, db, 0.5)[0])
    assert loose >= strict


# ---- motion: ref:tests/test_motion_estimator.cpp ---------------------------------------------
def _rot_err_deg(R1, R2):
    c = (np.trace(R1.T @ R2) - 1) / 2
    return np.degrees(np.arccos(np.clip(c, -1, 1)))


# (the _zyx files hold the points a g++ build of the reference's test draws, the others a clang build's)
@pytest.mark.parametrize("name", ["forward", "rotation", "forward_zyx", "rotation_zyx"])
def test_reference_motion_scenes(oracle, name):
    z = np.load(os.path.join(GOLD, f"motion_scene_{name}.npz"))
    r = oracle.estimate_motion(z["p1"], z["p2"], KITTI_K)
    assert r["valid"] and r["inliers"] > 50
    if name.startswith("forward"):
        assert r["t"][2] > 0.9
    # stronger than the reference: the known ground truth is recovered
    assert _rot_err_deg(r["R"], z["R"]) < 0.01
    assert np.linalg.norm(r["t"] - z["t"]) < 1e-4
    assert abs(np.linalg.norm(r["t"]) - 1) < 1e-12
    assert abs(np.linalg.det(r["R"]) - 1) < 1e-9


def test_motion_fixture_draw_orders_are_the_two_compilers(tmp_path):
    """ref:tests/test_motion_estimator.cpp:39-40 passes dist_x(rng), dist_y(rng), dist_z(rng) as the
    arguments of one constructor call; the order in which they are evaluated is unspecified.  A g++
    build of that expression (compiled here) gives the first point of the _zyx fixtures, a
    brace-initialised (left-to-right by rule) one the first point of the plain fixtures."""
    import shutil
    import subprocess
    if shutil.which("g++") is None:
        pytest.skip("g++ not available")
    src = tmp_path / "draw.cpp"
    src.write_text("""
#include <cstdio>
#include <random>
struct V { double x, y, z; V(double a, double b, double c) : x(a), y(b), z(c) {} };
int main() {
    std::mt19937 r1(42), r2(42);
    std::uniform_real_distribution<double> dx(100, 1100), dy(50, 300), dz(5, 50);
    V call(dx(r1), dy(r1), dz(r1));      // unspecified order
    V brace{dx(r2), dy(r2), dz(r2)};     // left to right
    std::printf("%.17g %.17g %.17g %.17g %.17g %.17g\\n", call.x, call.y, call.z, brace.x, brace.y, brace.z);
}
""")
    exe = tmp_path / "draw"
    subprocess.check_call(["g++", "-O0", "-o", str(exe), str(src)])
    v = [float(x) for x in subprocess.check_output([str(exe)]).split()]
    plain = np.load(os.path.join(GOLD, "motion_scene_forward.npz"))["p1"][0]
    zyx = np.load(os.path.join(GOLD, "motion_scene_forward_zyx.npz"))["p1"][0]
    # the first image point of a scene is the drawn (x, y) itself, rounded to float
    assert np.allclose(v[3:5], plain, rtol=1e-6) and np.allclose(v[0:2], zyx, rtol=1e-6)
    assert abs(plain[0] - 896.54297) < 1e-3 and abs(zyx[0] - 879.691) < 1e-3


def _pattern_table(path):
    import re
    txt = re.sub(r"/\*.*?\*/", "", open(path).read(), flags=re.S)
    return np.array([int(v) for v in re.findall(r"-?\d+", txt)], np.int64)


def test_rbrief_pattern_is_pinned():
    """The 256 x 4 rBRIEF test table is the one piece of OpenCV's ORB that a third party publishes as
    data: scikit-image ships its copy of bit_pattern_31_ as orb_descriptor_positions.txt.  Where that
    file exists (skimage itself need not be importable) the table is compared entry for entry;
    everywhere, a digest taken when that comparison held pins the table against later edits, and the
    oracle's and the GPU library's copies must be the same."""
    import glob
    import hashlib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    ours = _pattern_table(os.path.join(root, "oracle", "orb_pattern.inc"))
    gpu = _pattern_table(os.path.join(root, "visual_odometry_amd", "csrc", "orb_pattern.inc"))
    assert ours.shape == (1024,) and np.array_equal(ours, gpu)
    assert hashlib.sha256(ours.astype("<i1").tobytes()).hexdigest() == \
        "2164181aea6ff9ac426ca512d5130d15e1f6e3cd47b1cbdd568bbe1e55d49023"
    found = glob.glob("/opt/conda/lib/python3*/site-packages/skimage/feature/orb_descriptor_positions.txt") + \
        glob.glob("/usr/lib/python3*/site-packages/skimage/feature/orb_descriptor_positions.txt")
    if not found:
        pytest.skip("no scikit-image data file here: the digest above is the pin")
    sk = np.loadtxt(found[0])
    assert sk.shape == (256, 4) and np.array_equal(sk, np.round(sk))
    assert np.array_equal(ours.reshape(256, 4), sk.astype(np.int64))


def test_too_few_points_invalid(oracle):
    p1 = np.array([[100, 100], [200, 200], [300, 300]], np.float32)
    p2 = p1 + np.array([10, 0], np.float32)
    r = oracle.estimate_motion(p1, p2, KITTI_K)
    assert not r["valid"] and r["inliers"] == 0


def test_five_point_contains_true_essential(oracle):
    rng = np.random.default_rng(5)
    for trial in range(20):
        a = rng.normal(size=3) * 0.2
        th = np.linalg.norm(a)
        k = a / th
        Kx = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
        R = np.eye(3) + np.sin(th) * Kx + (1 - np.cos(th)) * Kx @ Kx
        t = rng.normal(size=3)
        t /= np.linalg.norm(t)
        X = np.column_stack([rng.uniform(-2, 2, 5), rng.uniform(-2, 2, 5), rng.uniform(4, 9, 5)])
        x1 = X[:, :2] / X[:, 2:]
        Y = X @ R.T + t
        x2 = Y[:, :2] / Y[:, 2:]
        Es = oracle.five_point(x1, x2)
        assert 1 <= len(Es) <= 10
        tx = np.array([[0, -t[2], t[1]], [t[2], 0, -t[0]], [-t[1], t[0], 0]])
        Et = tx @ R
        Et /= np.linalg.norm(Et)
        best = min(min(np.linalg.norm(E - Et), np.linalg.norm(E + Et)) for E in Es)
        assert best < 1e-6
        for E in Es:                                     # every model satisfies the five constraints
            r = [np.array([*x2[i], 1]) @ E @ np.array([*x1[i], 1]) for i in range(5)]
            assert np.max(np.abs(r)) < 1e-8
            assert abs(np.linalg.det(E)) < 1e-8


def test_rng_stream_first_values(oracle):
    # cv::RNG(-1): state = (uint32)state * 4164903690 + (state >> 32); hand-rolled check
    state = (1 << 64) - 1
    vals = []
    for _ in range(5):
        state = ((state & 0xFFFFFFFF) * 4164903690 + (state >> 32)) & ((1 << 64) - 1)
        vals.append((state & 0xFFFFFFFF) % 100)
    # five distinct draws are the first sample unless a duplicate forces a redraw
    p = np.random.default_rng(1).uniform(0, 500, (100, 2)).astype(np.float32)
    _, _, _, iters, samples = oracle.find_essential(p, p + 1, KITTI_K, fixed_iters=1)
    if len(set(vals)) == 5:
        assert list(samples[0]) == vals
