"""ctypes binding of libvo_hip.so (the C ABI declared in include/vo_hip.h).

There is no CPU fallback: if the library is missing or a call fails, an
exception is raised.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libvo_hip.so")

KP_DTYPE = np.dtype(
    [("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"), ("response", "<f4"),
     ("octave", "<i4"), ("class_id", "<i4")]
)
PAIR_DTYPE = np.dtype(
    [("R", "<f8", (9,)), ("t", "<f8", (3,)), ("inliers", "<i4"), ("valid", "<i4"),
     ("n_kp1", "<i4"), ("n_kp2", "<i4"), ("n_matches", "<i4"), ("ransac_iters", "<i4"),
     ("status", "<i4"), ("models_scored", "<i4")]
)

EXPORTS = [
    "vo_device_count", "vo_ctx_create", "vo_ctx_destroy", "vo_last_error", "vo_ctx_stream", "vo_sync", "vo_set_timing",
    "vo_kernel_time_ms", "vo_ctx_set_option", "vo_ctx_get_option", "vo_orb_detect", "vo_orb_detect_batch_dev", "vo_hamming_knn2",
    "vo_hamming_knn2_dev", "vo_hamming_knn2_multi", "vo_hamming_knn2_multi_dev", "vo_match_ratio", "vo_match_ratio_static", "vo_estimate_motion", "vo_find_essential",
    "vo_match_images", "vo_pipeline_pairs_dev", "vo_pipeline_sequence_dev", "vo_pipeline_ticket", "vo_pipeline_wait", "vo_pipeline_sequence", "vo_pipeline_sequence_multi", "vo_block_range", "vo_debug_level", "vo_debug_candidates", "vo_debug_fast_stats", "vo_debug_fast_thresholds", "vo_debug_five_point",
]

TRUNC_SCORE_LIST, TRUNC_KEYPOINT_LIST, TRUNC_OUTPUT = 1, 2, 4   # vo_pair_result.status bits (VO_TRUNC_*)
WAIT_INPUTS, WAIT_RESULTS = 1, 2                                 # vo_pipeline_wait (VO_WAIT_*)
# vo_ctx_set_option / vo_ctx_get_option (VO_OPT_*)
OPTIONS = {"five_point_literal": 0, "fast_raise": 1, "knn_mode": 2, "knn_waves": 3, "knn_queries_per_lane": 4,
           "second_chart": 5, "knn_train_split": 6, "solve_variant": 7, "pipeline_graph": 8}
T_GROUPS = ["resize", "fast", "select", "blur", "describe", "knn2", "ratio", "ransac",
            "ransac_solve", "ransac_score"]          # the last two split the RANSAC chain per round
T_TOP = T_GROUPS[:8]

_lib = None


class VoError(RuntimeError):
    pass


def load():
    """Loads libvo_hip.so and declares its prototypes."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise VoError(f"{LIB_PATH} is missing: build it with `make` (hipcc, gfx950)")
    # PyTorch wheels carry their own libamdhip64.so.7.  Whichever copy is mapped first
    # serves the whole process, and torch only finds the GPU when it is its own copy, so
    # torch (used by callers for device buffers and torch.distributed) goes first.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(LIB_PATH)
    vp, i, f, d, sz = C.c_void_p, C.c_int, C.c_float, C.c_double, C.c_size_t
    lib.vo_ctx_create.argtypes = [i, C.POINTER(vp)]
    lib.vo_ctx_destroy.argtypes = [vp]
    lib.vo_ctx_destroy.restype = None
    lib.vo_last_error.argtypes = [vp]
    lib.vo_last_error.restype = C.c_char_p
    lib.vo_ctx_stream.argtypes = [vp]
    lib.vo_ctx_stream.restype = vp
    lib.vo_sync.argtypes = [vp]
    lib.vo_pipeline_ticket.argtypes = [vp]
    lib.vo_pipeline_ticket.restype = C.c_ulonglong
    lib.vo_pipeline_wait.argtypes = [vp, C.c_ulonglong, i]
    lib.vo_set_timing.argtypes = [vp, i]
    lib.vo_kernel_time_ms.argtypes = [vp, i, C.POINTER(i)]
    lib.vo_kernel_time_ms.restype = f
    lib.vo_ctx_set_option.argtypes = [vp, i, i]
    lib.vo_ctx_get_option.argtypes = [vp, i, C.POINTER(i)]
    lib.vo_orb_detect.argtypes = [vp, vp, i, i, i, i, vp, vp, i, C.POINTER(i)]
    lib.vo_orb_detect_batch_dev.argtypes = [vp, vp, i, i, i, i, sz, i, vp, vp, i, vp]
    lib.vo_hamming_knn2.argtypes = [vp, vp, i, vp, i, vp, vp]
    lib.vo_hamming_knn2_dev.argtypes = [vp, vp, i, vp, i, vp, vp]
    lib.vo_hamming_knn2_multi.argtypes = [vp, i, vp, i, vp, i, vp, vp]
    lib.vo_hamming_knn2_multi_dev.argtypes = [vp, i, vp, i, vp, i, vp, vp]
    lib.vo_match_ratio.argtypes = [vp, vp, i, vp, i, f, vp, vp, vp, C.POINTER(i)]
    lib.vo_match_ratio_static.argtypes = [vp, vp, i, vp, i, f, vp, vp, vp, vp]
    lib.vo_estimate_motion.argtypes = [vp, vp, vp, i, vp, d, d, i, vp, vp, C.POINTER(i), C.POINTER(i), vp]
    lib.vo_find_essential.argtypes = [vp, vp, vp, i, vp, d, d, i, vp, vp, C.POINTER(i), C.POINTER(i), vp]
    lib.vo_match_images.argtypes = [vp, vp, vp, i, i, i, i, f, vp, C.POINTER(i), vp, C.POINTER(i), i, vp, vp, vp,
                                    C.POINTER(i)]
    lib.vo_pipeline_pairs_dev.argtypes = [vp, vp, i, i, i, i, sz, i, f, vp, d, d, i, vp]
    lib.vo_pipeline_sequence_dev.argtypes = [vp, vp, i, i, i, i, sz, i, f, vp, d, d, i, vp]
    lib.vo_pipeline_sequence.argtypes = [vp, vp, i, i, i, i, sz, i, f, vp, d, d, i, vp]
    lib.vo_pipeline_sequence_multi.argtypes = [vp, i, vp, i, i, i, i, sz, i, f, vp, d, d, i, vp]
    lib.vo_block_range.argtypes = [i, i, i, C.POINTER(i), C.POINTER(i)]
    lib.vo_debug_level.argtypes = [vp, i, i, i, vp, sz, C.POINTER(i), C.POINTER(i)]
    lib.vo_debug_candidates.argtypes = [vp, i, i, vp, i, C.POINTER(i)]
    lib.vo_debug_fast_stats.argtypes = [vp, i, vp]
    lib.vo_debug_fast_thresholds.argtypes = [vp, i, vp, C.POINTER(i)]
    lib.vo_debug_five_point.argtypes = [vp, vp, vp, i, vp, vp]
    _lib = lib
    return lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


class Context:
    """One GPU context (stream + workspace).  Not thread-safe."""

    def __init__(self, device=0):
        self.lib = load()
        h = C.c_void_p()
        rc = self.lib.vo_ctx_create(int(device), C.byref(h))
        if rc != 0 or not h:
            raise VoError(f"vo_ctx_create(device={device}) failed with code {rc}")
        self.h = h
        self.device = device

    def close(self):
        if getattr(self, "h", None):
            self.lib.vo_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, what, allow_truncated=False):
        if rc == 3 and allow_truncated:   # VO_ERR_CAPACITY from a host-buffer call: the outputs are still written
            return
        if rc != 0:
            raise VoError(f"{what} failed (code {rc}): {self.lib.vo_last_error(self.h).decode()}")

    @property
    def stream(self):
        return self.lib.vo_ctx_stream(self.h)

    def sync(self):
        self._check(self.lib.vo_sync(self.h), "vo_sync")

    def pipeline_ticket(self):
        """Ticket of the device-resident pipeline call enqueued last (0: none yet)."""
        return int(self.lib.vo_pipeline_ticket(self.h))

    def pipeline_wait(self, ticket, inputs=False, results=False):
        """Waits until the call with that ticket has read its images for the last time and / or
        written its result records, without waiting for calls enqueued after it."""
        self._check(self.lib.vo_pipeline_wait(self.h, int(ticket), (WAIT_INPUTS if inputs else 0) | (WAIT_RESULTS if results else 0)),
                    "vo_pipeline_wait")

    def set_option(self, name, value):
        """Sets one of the per-context switches (OPTIONS; include/vo_hip.h VO_OPT_*)."""
        self._check(self.lib.vo_ctx_set_option(self.h, OPTIONS[name], int(value)), "vo_ctx_set_option")

    def get_option(self, name):
        v = C.c_int()
        self._check(self.lib.vo_ctx_get_option(self.h, OPTIONS[name], C.byref(v)), "vo_ctx_get_option")
        return v.value

    def set_timing(self, on=True, groups=None):
        """Brackets the launches of the named kernel groups (default: all) with HIP events."""
        mask = 0
        if on:
            mask = (1 << len(T_TOP)) - 1 if groups is None else sum(1 << T_GROUPS.index(g) for g in groups)
        self.lib.vo_set_timing(self.h, mask)

    def kernel_time_ms(self, group):
        """(mean ms per launch, launches) of a kernel group since timing was enabled."""
        n = C.c_int()
        ms = float(self.lib.vo_kernel_time_ms(self.h, group, C.byref(n)))
        return ms, n.value

    # ---- host-buffer entry points ------------------------------------------------
    def orb_detect(self, img, max_features=2000, want_desc=True, cap=None, allow_truncated=False):
        if img is None or img.size == 0:
            n = C.c_int(-1)
            self._check(self.lib.vo_orb_detect(self.h, None, 0, 0, 0, max_features, None, None, 1, C.byref(n)),
                        "vo_orb_detect")
            return np.zeros(0, KP_DTYPE), (np.zeros((0, 32), np.uint8) if want_desc else None)
        img = np.ascontiguousarray(img, np.uint8)
        rows, cols = img.shape
        cap = cap or max_features + 8 * 256
        kps = np.zeros(cap, KP_DTYPE)
        desc = np.zeros((cap, 32), np.uint8) if want_desc else None
        n = C.c_int()
        self._check(self.lib.vo_orb_detect(self.h, _p(img), rows, cols, img.strides[0], int(max_features), _p(kps),
                                           _p(desc), cap, C.byref(n)), "vo_orb_detect", allow_truncated)
        return kps[: n.value].copy(), (desc[: n.value].copy() if want_desc else None)

    def debug_level(self, image, level, blurred=False):
        w, h = C.c_int(), C.c_int()
        # a first call with a zero capacity only reports the level's size
        self.lib.vo_debug_level(self.h, image, level, int(blurred), None, 0, C.byref(w), C.byref(h))
        out = np.zeros(max(w.value * h.value, 1), np.uint8)
        self._check(self.lib.vo_debug_level(self.h, image, level, int(blurred), _p(out), out.size, C.byref(w),
                                            C.byref(h)), "vo_debug_level")
        return out[: w.value * h.value].reshape(h.value, w.value).copy()

    def fast_thresholds(self, image):
        """None if the last detection ran every FAST tile with the plain threshold, else
        (thresholds[8], redo[8]) of one image."""
        out, raised = np.zeros(16, np.int32), C.c_int()
        self._check(self.lib.vo_debug_fast_thresholds(self.h, image, _p(out), C.byref(raised)), "vo_debug_fast_thresholds")
        return (out[:8].copy(), out[8:].copy()) if raised.value else None

    def fast_stats(self, enable):
        """enable=True: start counting; False: stop and return dict(pixels, pretest_pass, corners, candidates)."""
        out = np.zeros(4, np.uint64)
        self._check(self.lib.vo_debug_fast_stats(self.h, int(bool(enable)), _p(out)), "vo_debug_fast_stats")
        return dict(zip(("pixels", "pretest_pass", "corners", "candidates"), (int(v) for v in out)))

    def debug_candidates(self, image, level, cap=1 << 20):
        out = np.zeros((cap, 3), np.int32)
        n = C.c_int()
        self._check(self.lib.vo_debug_candidates(self.h, image, level, _p(out), cap, C.byref(n)),
                    "vo_debug_candidates")
        return out[: n.value].copy()

    def hamming_knn2(self, q, t):
        q = np.ascontiguousarray(q, np.uint8).reshape(-1, 32)
        t = np.ascontiguousarray(t, np.uint8).reshape(-1, 32)
        idx = np.full((len(q), 2), -7, np.int32)
        dist = np.full((len(q), 2), -7, np.int32)
        self._check(self.lib.vo_hamming_knn2(self.h, _p(q), len(q), _p(t), len(t), _p(idx), _p(dist)),
                    "vo_hamming_knn2")
        return idx, dist

    def match_ratio(self, q, t, ratio=0.75):
        q = np.ascontiguousarray(q, np.uint8).reshape(-1, 32)
        t = np.ascontiguousarray(t, np.uint8).reshape(-1, 32)
        n = max(len(q), 1)
        qi, ti, d = np.zeros(n, np.int32), np.zeros(n, np.int32), np.zeros(n, np.float32)
        m = C.c_int()
        self._check(self.lib.vo_match_ratio(self.h, _p(q), len(q), _p(t), len(t), float(ratio), _p(qi), _p(ti),
                                            _p(d), C.byref(m)), "vo_match_ratio")
        return qi[: m.value].copy(), ti[: m.value].copy(), d[: m.value].copy()

    def match_images(self, img1, img2, max_features=2000, ratio=0.75, allow_truncated=False):
        """match_images_orb for one pair of host images: (kps1, kps2, query_idx, train_idx, distance)."""
        img1 = np.ascontiguousarray(img1, np.uint8)
        img2 = np.ascontiguousarray(img2, np.uint8)
        if img1.shape != img2.shape:
            raise ValueError("the two images must have the same shape")
        rows, cols = img1.shape
        cap = max_features + 8 * 256
        bufs = getattr(self, "_mi_bufs", None)
        if bufs is None or bufs[0] != cap:
            bufs = (cap, np.zeros(cap, KP_DTYPE), np.zeros(cap, KP_DTYPE), np.zeros(cap, np.int32),
                    np.zeros(cap, np.int32), np.zeros(cap, np.float32))
            self._mi_bufs = bufs
        _, k1, k2, qi, ti, d = bufs
        n1, n2, m = C.c_int(), C.c_int(), C.c_int()
        self._check(self.lib.vo_match_images(self.h, _p(img1), _p(img2), rows, cols, img1.strides[0], int(max_features),
                                             float(ratio), _p(k1), C.byref(n1), _p(k2), C.byref(n2), cap, _p(qi),
                                             _p(ti), _p(d), C.byref(m)), "vo_match_images", allow_truncated)
        return (k1[: n1.value].copy(), k2[: n2.value].copy(), qi[: m.value].copy(), ti[: m.value].copy(),
                d[: m.value].copy())

    def estimate_motion(self, p1, p2, K, threshold=1.0, confidence=0.999, fixed_iters=0):
        p1 = np.ascontiguousarray(p1, np.float32).reshape(-1, 2)
        p2 = np.ascontiguousarray(p2, np.float32).reshape(-1, 2)
        n = len(p1)
        Kd = np.asarray(K, np.float64)
        R, t = np.zeros((3, 3)), np.zeros(3)
        inl, valid = C.c_int(), C.c_int()
        mask = np.zeros(max(n, 1), np.uint8)
        self._check(self.lib.vo_estimate_motion(self.h, _p(p1), _p(p2), n, _p(Kd), threshold, confidence,
                                                int(fixed_iters), _p(R), _p(t), C.byref(inl), C.byref(valid),
                                                _p(mask)), "vo_estimate_motion")
        return dict(valid=bool(valid.value), inliers=inl.value, R=R, t=t, mask=mask[:n])

    def find_essential(self, p1, p2, K, threshold=1.0, confidence=0.999, fixed_iters=0):
        p1 = np.ascontiguousarray(p1, np.float32).reshape(-1, 2)
        p2 = np.ascontiguousarray(p2, np.float32).reshape(-1, 2)
        n = len(p1)
        Kd = np.asarray(K, np.float64)
        E = np.zeros((3, 3))
        mask = np.zeros(max(n, 1), np.uint8)
        good, iters = C.c_int(), C.c_int()
        samples = np.full((max(1000, fixed_iters), 5), -1, np.int32)
        self._check(self.lib.vo_find_essential(self.h, _p(p1), _p(p2), n, _p(Kd), threshold, confidence,
                                               int(fixed_iters), _p(E), _p(mask), C.byref(good), C.byref(iters),
                                               _p(samples)), "vo_find_essential")
        return good.value, E, mask[:n], iters.value, samples[: iters.value]

    def five_point(self, x1, x2):
        """x1, x2: (count, 5, 2) normalised points.  Returns a list of (k, 3, 3) arrays."""
        x1 = np.ascontiguousarray(x1, np.float64).reshape(-1, 5, 2)
        x2 = np.ascontiguousarray(x2, np.float64).reshape(-1, 5, 2)
        cnt = len(x1)
        Es = np.zeros((cnt, 10, 3, 3))
        nm = np.zeros(cnt, np.int32)
        self._check(self.lib.vo_debug_five_point(self.h, _p(x1), _p(x2), cnt, _p(Es), _p(nm)), "vo_debug_five_point")
        return [Es[i, : nm[i]].copy() for i in range(cnt)]

    # ---- device-resident entry points (pointers are integers / c_void_p) ---------
    def orb_detect_batch_dev(self, d_imgs, n_images, rows, cols, step, image_stride, max_features, d_kps, d_desc,
                             cap, d_counts):
        self._check(self.lib.vo_orb_detect_batch_dev(self.h, d_imgs, n_images, rows, cols, step, image_stride,
                                                     max_features, d_kps, d_desc, cap, d_counts),
                    "vo_orb_detect_batch_dev")

    def hamming_knn2_dev(self, d_q, nq, d_t, nt, d_idx2, d_dist2):
        self._check(self.lib.vo_hamming_knn2_dev(self.h, d_q, nq, d_t, nt, d_idx2, d_dist2), "vo_hamming_knn2_dev")

    def match_ratio_static(self, d_q, nq, d_t, nt, ratio, d_query_idx, d_train_idx, d_distance, d_n_out):
        """vo_match_ratio on device buffers with the library's static workspace (allocates nothing)."""
        self._check(self.lib.vo_match_ratio_static(self.h, d_q, nq, d_t, nt, ratio, d_query_idx, d_train_idx,
                                                   d_distance, d_n_out), "vo_match_ratio_static")

    def pipeline_pairs_dev(self, d_imgs, n_pairs, rows, cols, step, image_stride, max_features, ratio, K, threshold,
                           confidence, fixed_iters, d_results):
        Kd = np.asarray(K, np.float64)
        self._check(self.lib.vo_pipeline_pairs_dev(self.h, d_imgs, n_pairs, rows, cols, step, image_stride,
                                                   max_features, ratio, _p(Kd), threshold, confidence, fixed_iters,
                                                   d_results), "vo_pipeline_pairs_dev")

    def pipeline_sequence_dev(self, d_imgs, n_frames, rows, cols, step, image_stride, max_features, ratio, K,
                              threshold, confidence, fixed_iters, d_results):
        """n_frames consecutive frames -> the n_frames - 1 pairs (i, i+1), each frame detected once."""
        Kd = np.asarray(K, np.float64)
        self._check(self.lib.vo_pipeline_sequence_dev(self.h, d_imgs, n_frames, rows, cols, step, image_stride,
                                                      max_features, ratio, _p(Kd), threshold, confidence,
                                                      fixed_iters, d_results), "vo_pipeline_sequence_dev")

    def pipeline_sequence(self, frames, max_features, K, ratio=0.75, threshold=1.0, confidence=0.999, fixed_iters=0):
        """Host arrays in (n_frames x H x W uint8), PAIR_DTYPE records of the n_frames - 1 pairs out."""
        frames = np.ascontiguousarray(frames, np.uint8)
        n, h, w = frames.shape
        Kd = np.asarray(K, np.float64)
        out = np.zeros(n - 1, PAIR_DTYPE)
        self._check(self.lib.vo_pipeline_sequence(self.h, _p(frames), n, h, w, w, h * w, max_features, ratio, _p(Kd),
                                                  threshold, confidence, fixed_iters, _p(out)), "vo_pipeline_sequence")
        return out


def block_range(n, part, parts):
    """[begin, end) of n units owned by part `part` of `parts` (vo_block_range; no GPU call)."""
    b, e = C.c_int(), C.c_int()
    if load().vo_block_range(int(n), int(part), int(parts), C.byref(b), C.byref(e)) != 0:
        raise VoError("vo_block_range: bad argument")
    return b.value, e.value


def pipeline_sequence_multi(contexts, frames, max_features, K, ratio=0.75, threshold=1.0, confidence=0.999,
                            fixed_iters=0):
    """vo_pipeline_sequence over several contexts (one per GPU): host frames in, PAIR_DTYPE records of the
    n_frames - 1 pairs out, in pair order."""
    frames = np.ascontiguousarray(frames, np.uint8)
    n, h, w = frames.shape
    Kd = np.asarray(K, np.float64)
    out = np.zeros(n - 1, PAIR_DTYPE)
    arr = (C.c_void_p * len(contexts))(*[c.h for c in contexts])
    rc = load().vo_pipeline_sequence_multi(arr, len(contexts), _p(frames), n, h, w, w, h * w, max_features, ratio,
                                           _p(Kd), threshold, confidence, fixed_iters, _p(out))
    contexts[0]._check(rc, "vo_pipeline_sequence_multi")
    return out


def hamming_knn2_multi(contexts, q, t):
    """vo_hamming_knn2 over several contexts (one per GPU): query rows sharded in contiguous blocks,
    train set replicated, host arrays in and out."""
    q = np.ascontiguousarray(q, np.uint8).reshape(-1, 32)
    t = np.ascontiguousarray(t, np.uint8).reshape(-1, 32)
    idx = np.full((len(q), 2), -7, np.int32)
    dist = np.full((len(q), 2), -7, np.int32)
    arr = (C.c_void_p * len(contexts))(*[c.h for c in contexts])
    rc = load().vo_hamming_knn2_multi(arr, len(contexts), _p(q), len(q), _p(t), len(t), _p(idx), _p(dist))
    contexts[0]._check(rc, "vo_hamming_knn2_multi")
    return idx, dist


def hamming_knn2_multi_dev(contexts, d_q_blocks, nq, d_t_copies, nt, d_idx2, d_dist2):
    """Device-resident form: d_q_blocks[k] / d_t_copies[k] are device addresses on context k's GPU,
    the tables live on context 0's GPU.  Only enqueues: sync every context afterwards."""
    n = len(contexts)
    arr = (C.c_void_p * n)(*[c.h for c in contexts])
    qs = (C.c_void_p * n)(*d_q_blocks)
    ts = (C.c_void_p * n)(*d_t_copies)
    rc = load().vo_hamming_knn2_multi_dev(arr, n, qs, int(nq), ts, int(nt), d_idx2, d_dist2)
    contexts[0]._check(rc, "vo_hamming_knn2_multi_dev")
