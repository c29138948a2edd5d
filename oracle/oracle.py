"""ctypes front end of the CPU oracle (oracle/_build/libvo_oracle.so).

Test infrastructure: imported by tests/, by ``__graft_entry__.smoke()`` and by
the ``cpu_baseline`` leg of ``bench.py`` only.  The shipped GPU path never
imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libvo_oracle.so")

KP_DTYPE = np.dtype(
    [("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"), ("response", "<f4"),
     ("octave", "<i4"), ("class_id", "<i4")]
)

_lib = None


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = C.CDLL(_SO)
        _lib.oracle_fast_atan2.restype = C.c_float
        _lib.oracle_fast_atan2.argtypes = [C.c_float, C.c_float]
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def level_geometry(rows, cols, level):
    w, h, s = C.c_int(), C.c_int(), C.c_float()
    lib().oracle_level_geometry(rows, cols, level, C.byref(w), C.byref(h), C.byref(s))
    return w.value, h.value, s.value


def pyramid_level(img, level, blurred=False):
    img = np.ascontiguousarray(img, dtype=np.uint8)
    rows, cols = img.shape
    w, h, _ = level_geometry(rows, cols, level)
    out = np.empty((h, w), np.uint8)
    lib().oracle_pyramid_level(_p(img), rows, cols, cols, level, int(blurred), _p(out))
    return out


def fast(img, threshold=20):
    img = np.ascontiguousarray(img, dtype=np.uint8)
    rows, cols = img.shape
    cap = rows * cols // 4 + 16
    out = np.empty((cap, 3), np.int32)
    n = lib().oracle_fast(_p(img), rows, cols, cols, threshold, _p(out), cap)
    return out[:n].copy()


def features_per_level(nfeatures):
    out = np.zeros(8, np.int32)
    lib().oracle_features_per_level(int(nfeatures), _p(out))
    return out


def fast_atan2(y, x):
    return lib().oracle_fast_atan2(float(y), float(x))


def orb_detect(img, max_features=2000, want_desc=True, order_mode=0):
    """Returns (keypoints[KP_DTYPE], descriptors[n,32] or None)."""
    if img is None or img.size == 0:
        return np.zeros(0, KP_DTYPE), (np.zeros((0, 32), np.uint8) if want_desc else None)
    img = np.ascontiguousarray(img, dtype=np.uint8)
    rows, cols = img.shape
    cap = rows * cols // 4 + 16
    kps = np.zeros(cap, KP_DTYPE)
    desc = np.zeros((cap, 32), np.uint8)
    n = lib().oracle_orb_detect(_p(img), rows, cols, cols, int(max_features), int(want_desc),
                                int(order_mode), _p(kps), _p(desc), cap)
    return kps[:n].copy(), (desc[:n].copy() if want_desc else None)


def knn2(q, t):
    q = np.ascontiguousarray(q, np.uint8).reshape(-1, 32)
    t = np.ascontiguousarray(t, np.uint8).reshape(-1, 32)
    idx = np.empty((len(q), 2), np.int32)
    dist = np.empty((len(q), 2), np.int32)
    lib().oracle_knn2(_p(q), len(q), _p(t), len(t), _p(idx), _p(dist))
    return idx, dist


def match(q, t, ratio=0.75):
    q = np.ascontiguousarray(q, np.uint8).reshape(-1, 32)
    t = np.ascontiguousarray(t, np.uint8).reshape(-1, 32)
    n = len(q)
    qi = np.empty(max(n, 1), np.int32)
    ti = np.empty(max(n, 1), np.int32)
    d = np.empty(max(n, 1), np.float32)
    m = lib().oracle_match(_p(q), n, _p(t), len(t), C.c_float(ratio), _p(qi), _p(ti), _p(d))
    return qi[:m].copy(), ti[:m].copy(), d[:m].copy()


def set_five_point_charts(n):
    """Null-space charts the minimal solver is run in (default 4: complete solution sets for
    parity checks; 1: OpenCV's single chart, for timing the CPU baseline)."""
    lib().oracle_set_five_point_charts(int(n))


def set_variant(trig_float=False, five_point_literal=False):
    """Study switches: descriptor rotation through libm's cosf / sinf instead of the double routines
    rounded to float; the minimal solver as OpenCV writes it (one chart, roots with |imag| <= 1e-10,
    no refinement, no duplicate drop).  Both off by default - the variant the GPU is pinned to."""
    lib().oracle_set_variant(int(trig_float), int(five_point_literal))


def five_point(x1, x2):
    x1 = np.ascontiguousarray(x1, np.float64).reshape(5, 2)
    x2 = np.ascontiguousarray(x2, np.float64).reshape(5, 2)
    Es = np.zeros((10, 3, 3))
    n = lib().oracle_five_point(_p(x1), _p(x2), _p(Es))
    return Es[:n].copy()


def find_essential(p1, p2, K, threshold=1.0, confidence=0.999, max_iters=1000, fixed_iters=0):
    """K = (fx, fy, cx, cy).  Returns (good, E, mask, iters_run, samples)."""
    p1 = np.ascontiguousarray(p1, np.float32).reshape(-1, 2)
    p2 = np.ascontiguousarray(p2, np.float32).reshape(-1, 2)
    n = len(p1)
    Kd = np.asarray(K, np.float64)
    E = np.zeros((3, 3))
    mask = np.zeros(max(n, 1), np.uint8)
    iters = C.c_int()
    cap = max(max_iters, fixed_iters, 1)
    samples = np.full((cap, 5), -1, np.int32)
    good = lib().oracle_find_essential(_p(p1), _p(p2), n, _p(Kd), C.c_double(threshold),
                                       C.c_double(confidence), max_iters, fixed_iters, _p(E),
                                       _p(mask), C.byref(iters), _p(samples))
    return good, E, mask[:n], iters.value, samples[: iters.value]


def recover_pose(E, p1, p2, K, mask):
    p1 = np.ascontiguousarray(p1, np.float32).reshape(-1, 2)
    p2 = np.ascontiguousarray(p2, np.float32).reshape(-1, 2)
    E = np.ascontiguousarray(E, np.float64)
    Kd = np.asarray(K, np.float64)
    m = np.ascontiguousarray(mask, np.uint8).copy()
    R = np.zeros((3, 3))
    t = np.zeros(3)
    cnt = lib().oracle_recover_pose(_p(E), _p(p1), _p(p2), len(p1), _p(Kd), _p(m), _p(R), _p(t))
    return cnt, R, t, m


def estimate_motion(p1, p2, K, threshold=1.0, confidence=0.999, fixed_iters=0):
    """Returns dict(valid, inliers, R, t, mask) with estimate_motion's semantics."""
    p1 = np.ascontiguousarray(p1, np.float32).reshape(-1, 2)
    p2 = np.ascontiguousarray(p2, np.float32).reshape(-1, 2)
    n = len(p1)
    Kd = np.asarray(K, np.float64)
    R = np.zeros((3, 3))
    t = np.zeros(3)
    inl = C.c_int()
    mask = np.zeros(max(n, 1), np.uint8)
    valid = lib().oracle_estimate_motion(_p(p1), _p(p2), n, _p(Kd), C.c_double(threshold),
                                         C.c_double(confidence), fixed_iters, _p(R), _p(t),
                                         C.byref(inl), _p(mask))
    return dict(valid=bool(valid), inliers=inl.value, R=R, t=t, mask=mask[:n])
