"""Deterministic synthetic inputs for tests and benchmarks.

The datasets the reference is run on (TUM fr1_xyz, KITTI) cannot be fetched, so
frames are generated: a layered scene of textured rectangles at different
depths, seen by a camera that translates along +x between the two frames
(every layer shifts left by its own integer disparity, 12-40 px, so that
triangulated depths stay below recoverPose's 50-baseline cut-off) and also rotates
a little (an exact homography warp of the second frame), plus independent pixel
noise per frame.  The two views therefore obey a real two-view geometry
(R = I, t along x) and FAST fires densely at every pyramid level.
"""
import numpy as np


TUM_K = (517.3, 516.5, 318.6, 255.3)         # ref:data/tum/README.md:57-58
KITTI_K = (718.856, 718.856, 607.1928, 185.2157)  # ref:data/README.md:30-33


def default_K(width, height):
    """Intrinsics (fx, fy, cx, cy) used for a synthetic frame of this shape."""
    if (width, height) == (640, 480):
        return TUM_K
    if (width, height) == (1241, 376):
        return KITTI_K
    return (0.8 * width, 0.8 * width, width / 2.0, height / 2.0)


def _box3(img):
    """3x3 box filter with edge replication (stands in for lens / sensor blur)."""
    p = np.pad(img, 1, mode="edge")
    out = np.zeros_like(img)
    for dy in range(3):
        for dx in range(3):
            out += p[dy:dy + img.shape[0], dx:dx + img.shape[1]]
    return out / 9.0


def _texture(rng, h, w, sparse=False):
    """Blocky texture with corners at several scales: 16-px base cells plus 32-px and
    8-px ternary patterns, slightly blurred.  Dense: about a quarter of the pixels pass
    FAST's cardinal pre-test and every pyramid level over-fills its keypoint quota many
    times - far busier than indoor or road scenes.  sparse=True: 40-px cells with one
    80-px pattern, blurred twice - a few corners per thousand pixels, like a real frame."""
    base, cells, blurs = (40, ((80, 30.0),), 2) if sparse else (16, ((32, 40.0), (8, 40.0)), 1)
    img = rng.integers(40, 216, size=(h // base + 2, w // base + 2)).astype(np.float32)
    img = np.kron(img, np.ones((base, base), np.float32))[:h, :w]
    for cell, amp in cells:
        g = rng.integers(-1, 2, size=(h // cell + 2, w // cell + 2)).astype(np.float32)
        img += amp * np.kron(g, np.ones((cell, cell), np.float32))[:h, :w]
    for _ in range(blurs):
        img = _box3(img)
    return img


def _rotation(rx, ry, rz):
    cx, sx, cy, sy, cz, sz = np.cos(rx), np.sin(rx), np.cos(ry), np.sin(ry), np.cos(rz), np.sin(rz)
    Rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
    Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    Rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    return Rz @ Ry @ Rx


def _warp_rotation(img, K, R):
    """View of the same scene after rotating the camera by R (x_new = K R K^-1 x_old):
    exact for any depth, so it composes with the translational parallax. Bilinear."""
    h, w = img.shape
    fx, fy, cx, cy = K
    Km = np.array([[fx, 0, cx], [0, fy, cy], [0, 0, 1.0]])
    Hinv = Km @ R.T @ np.linalg.inv(Km)
    ys, xs = np.mgrid[0:h, 0:w].astype(np.float64)
    d = Hinv[2, 0] * xs + Hinv[2, 1] * ys + Hinv[2, 2]
    sx = np.clip((Hinv[0, 0] * xs + Hinv[0, 1] * ys + Hinv[0, 2]) / d, 0, w - 1.001)
    sy = np.clip((Hinv[1, 0] * xs + Hinv[1, 1] * ys + Hinv[1, 2]) / d, 0, h - 1.001)
    x0, y0 = np.floor(sx).astype(np.int64), np.floor(sy).astype(np.int64)
    ax, ay = sx - x0, sy - y0
    return ((1 - ay) * ((1 - ax) * img[y0, x0] + ax * img[y0, x0 + 1]) +
            ay * ((1 - ax) * img[y0 + 1, x0] + ax * img[y0 + 1, x0 + 1]))


def frame_pair(seed, height=480, width=640, n_layers=12, noise=3.0, K=None, rot_deg=(0.4, -0.7, 1.1), sparse=False,
               moving_fraction=0.0):
    """Returns (frame_a, frame_b) as uint8 arrays of shape (height, width).

    rot_deg: camera rotation (about x, y, z, degrees) between the frames, applied on
    top of the sideways translation; (0, 0, 0) leaves the exactly degenerate
    pure-translation geometry with horizontal integer disparities.
    sparse: low-texture scene (see _texture).  moving_fraction: share of the layers that
    move on their own (a vertical shift no camera motion explains): their matches are the
    outliers of the essential-matrix fit."""
    rng = np.random.default_rng(seed)
    pad = 48
    W = width + 2 * pad
    a = _texture(rng, height, W, sparse)
    b = np.roll(a, -12, axis=1)  # background: disparity 12 px
    for j in range(n_layers):
        h = int(rng.integers(height // 6, height // 2))
        w = int(rng.integers(width // 8, width // 3))
        y = int(rng.integers(0, height - h))
        x = int(rng.integers(pad, W - pad - w))
        d = int(rng.integers(14, 41))
        tex = _texture(rng, h, w, sparse)
        a[y:y + h, x:x + w] = tex
        if moving_fraction > 0 and rng.random() < moving_fraction:
            dy = int(rng.integers(6, 25)) * (1 if rng.random() < 0.5 else -1)
            y2 = min(max(y + dy, 0), height - h)
            b[y2:y2 + h, x - d:x - d + w] = tex
        else:
            b[y:y + h, x - d:x - d + w] = tex
    a = a[:, pad:pad + width]
    b = b[:, pad:pad + width]
    if any(rot_deg):
        Kc = K if K is not None else default_K(width, height)
        b = _warp_rotation(b, Kc, _rotation(*np.radians(rot_deg)))
    a = a + rng.normal(0.0, noise, a.shape)
    b = b + rng.normal(0.0, noise, b.shape)
    return (np.clip(np.rint(a), 0, 255).astype(np.uint8),
            np.clip(np.rint(b), 0, 255).astype(np.uint8))


def sequence(seed, n_frames, height=480, width=640, n_layers=12, noise=3.0, K=None, rot_amp_deg=(0.4, 0.7, 1.1)):
    """n_frames consecutive views of one layered scene from a camera that moves one baseline
    along +x per frame and wobbles a little: a stand-in for a TUM / KITTI directory.

    Every layer is a periodic texture behind a rectangular mask on a canvas wider than the
    view; frame k shows layer j rolled left by k * d_j pixels (d_j = its integer disparity per
    frame, 12-40 px, the background 12), nearer layers on top, then rotated by R_k (exact
    homography warp) and given independent pixel noise.  Consecutive frames therefore obey a
    real two-view geometry, like the two frames of frame_pair()."""
    rng = np.random.default_rng(seed)
    Wc = 2 * width                                  # canvas: masks re-enter far outside the view
    layers = [(12, _texture(rng, height, Wc), None)]
    for _ in range(n_layers):
        h = int(rng.integers(height // 6, height // 2))
        w = int(rng.integers(width // 8, width // 3))
        y = int(rng.integers(0, height - h))
        x = int(rng.integers(0, Wc - w))
        d = int(rng.integers(14, 41))
        mask = np.zeros((height, Wc), bool)
        mask[y:y + h, x:x + w] = True
        layers.append((d, _texture(rng, height, Wc), mask))
    layers.sort(key=lambda t: t[0])                 # far to near
    Kc = K if K is not None else default_K(width, height)
    frames = []
    for k in range(n_frames):
        view = None
        for d, tex, mask in layers:
            t = np.roll(tex, -k * d, axis=1)[:, :width]
            if mask is None:
                view = t.copy()
            else:
                m = np.roll(mask, -k * d, axis=1)[:, :width]
                view[m] = t[m]
        ang = np.radians([rot_amp_deg[0] * np.sin(0.30 * k), rot_amp_deg[1] * np.sin(0.21 * k + 1.0),
                          rot_amp_deg[2] * np.sin(0.17 * k + 2.0)])
        if np.any(ang != 0):
            view = _warp_rotation(view, Kc, _rotation(*ang))
        view = view + rng.normal(0.0, noise, view.shape)
        frames.append(np.clip(np.rint(view), 0, 255).astype(np.uint8))
    return frames


def checkerboard(height=480, width=640, cell=32, shift=0, noise_sigma=25.0, seed=0):
    """The reference's detector/matcher test fixture (a 0/255 checkerboard with
    additive Gaussian noise, saturated to uint8; ref:tests/test_feature_detector.cpp:11-23,
    ref:tests/test_feature_matcher.cpp:13-33) with a seeded numpy noise source in
    place of OpenCV's global RNG, whose stream is not reproducible here."""
    y, x = np.mgrid[0:height, 0:width]
    x2 = (x + shift) % width
    img = np.where(((x2 // cell) % 2) == ((y // cell) % 2), 255, 0).astype(np.float32)
    rng = np.random.default_rng(seed)
    # cv::randn into CV_8U saturates negative noise to 0, so only the positive
    # half is added (test_image_ += noise).
    noise = np.clip(np.rint(rng.normal(0.0, noise_sigma, img.shape)), 0, 255)
    return np.clip(img + noise, 0, 255).astype(np.uint8)


def descriptors(seed, n, dup_fraction=0.01):
    """n random 256-bit descriptors with a fraction of planted near-duplicates."""
    rng = np.random.default_rng(seed)
    d = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
    k = int(n * dup_fraction)
    if k > 0 and n > 1:
        src = rng.integers(0, n, size=k)
        dst = rng.integers(0, n, size=k)
        d[dst] = d[src]
        flip = rng.integers(0, 32, size=k)
        d[dst, flip] ^= (1 << rng.integers(0, 8, size=k)).astype(np.uint8)
    return d




_TWO_VIEW_KS = [(517.3, 516.5, 318.6, 255.3), (718.856, 718.856, 607.1928, 185.2157), (1400.0, 1390.0, 960.0, 540.0),
                (300.0, 320.0, 160.0, 120.0)]


def _random_rotation(rng, max_deg):
    v = rng.normal(size=3)
    v /= np.linalg.norm(v)
    a = np.deg2rad(rng.uniform(0, max_deg))
    Kx = np.array([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0]])
    return np.eye(3) + np.sin(a) * Kx + (1 - np.cos(a)) * Kx @ Kx


def two_view_scene(rng):
    """Random matched points of two views for RANSAC checks: (p1, p2, K, threshold, fixed_iters,
    (kind, noise, baseline)).  kind 0: general scene, 1: planar, 2: small rotation, 3: pure rotation
    (no translation), 4: a third of the matches are copies of the first.  5..3000 matches, pixel
    noise 0..2, 0..90 % gross outliers, rotation up to 20 degrees, baseline 0.02..1 at depth 4..12."""
    K = _TWO_VIEW_KS[rng.integers(len(_TWO_VIEW_KS))]
    n = int(rng.choice([5, 6, 8, 12, 20, 50, 120, 300, 700, 1500, 3000]))
    kind = rng.integers(5)
    X = np.stack([rng.uniform(-4, 4, n), rng.uniform(-3, 3, n), rng.uniform(4, 12, n)], 1)
    if kind == 1:
        X[:, 2] = 8.0 + 0.2 * X[:, 0]
    R = _random_rotation(rng, 20.0 if kind != 2 else 3.0)
    t = rng.normal(size=3)
    t *= (0.0 if kind == 3 else rng.choice([0.02, 0.3, 1.0])) / max(np.linalg.norm(t), 1e-9)
    Y = X @ R.T + t
    fx, fy, cx, cy = K
    p1 = np.stack([fx * X[:, 0] / X[:, 2] + cx, fy * X[:, 1] / X[:, 2] + cy], 1)
    p2 = np.stack([fx * Y[:, 0] / Y[:, 2] + cx, fy * Y[:, 1] / Y[:, 2] + cy], 1)
    sig = rng.choice([0.0, 0.3, 1.0, 2.0])
    p1 += rng.normal(0, sig, p1.shape)
    p2 += rng.normal(0, sig, p2.shape)
    out = rng.random(n) < rng.choice([0.0, 0.2, 0.5, 0.8, 0.9])
    p2[out] += rng.uniform(-150, 150, (int(out.sum()), 2))
    if kind == 4 and n > 10:
        p1[1::3] = p1[0]
        p2[1::3] = p2[0]
    thr = float(rng.choice([0.3, 1.0, 3.0]))
    fixed = int(rng.choice([0, 0, 0, 40, 600]))
    return p1.astype(np.float32), p2.astype(np.float32), K, thr, fixed, (int(kind), float(sig), float(np.linalg.norm(t)))
