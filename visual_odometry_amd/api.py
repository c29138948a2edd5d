"""Python face of the GPU path with the names the reference's compiled module binds
(ref:src/python_bindings.cpp:42-382): ``orb_matcher(max_features, ratio_threshold).match_images``,
``MatchResult.points1/points2/num_matches``, ``CameraIntrinsics``, ``MotionEstimatorConfig``,
``estimate_motion(points1, points2, intrinsics, config)``, ``MotionEstimate`` (truthy when valid),
``Pose``, ``Trajectory``, ``image_loader``.  Failures raise ``RuntimeError`` as the reference's
bindings do.  Everything runs through the C ABI (``capi.py``); ``process_sequence`` adds the
batched form the reference does not have."""
import os
import threading

import numpy as np

from . import capi

default_max_features = 2000
default_ratio_threshold = 0.75

_local = threading.local()


def _ctx():
    """One lazily created context per thread (device from VO_HIP_DEVICE, default 0)."""
    if getattr(_local, "ctx", None) is None:
        _local.ctx = capi.Context(int(os.environ.get("VO_HIP_DEVICE", "0")))
    return _local.ctx


def _gray_u8(img):
    a = np.asarray(img)
    if a.dtype != np.uint8 or a.ndim != 2:
        raise RuntimeError("expected a 2-D uint8 (grayscale) image")
    return np.ascontiguousarray(a)


class MatchResult:
    """Matched pixel coordinates (N x 2 float32 each) in query order, plus the raw index pairs."""

    def __init__(self, points1, points2, query_idx, train_idx, distances):
        self.points1, self.points2 = points1, points2
        self.query_idx, self.train_idx, self.distances = query_idx, train_idx, distances

    @property
    def num_matches(self):
        return len(self.points1)

    def __len__(self):
        return len(self.points1)


class orb_matcher:
    """ORB detect on both images + brute-force Hamming 2-NN + ratio test."""

    def __init__(self, max_features=default_max_features, ratio_threshold=default_ratio_threshold):
        self.max_features, self.ratio_threshold = int(max_features), float(ratio_threshold)

    @property
    def name(self):
        return "ORB"

    def match_images(self, img1, img2):
        empty = np.zeros((0, 2), np.float32)
        a, b = np.asarray(img1), np.asarray(img2)
        if a.size == 0 or b.size == 0:
            return MatchResult(empty, empty, np.zeros(0, np.int32), np.zeros(0, np.int32), np.zeros(0, np.float32))
        c = _ctx()
        k1, d1 = c.orb_detect(_gray_u8(a), self.max_features)
        k2, d2 = c.orb_detect(_gray_u8(b), self.max_features)
        if len(k1) == 0 or len(k2) == 0:
            return MatchResult(empty, empty, np.zeros(0, np.int32), np.zeros(0, np.int32), np.zeros(0, np.float32))
        qi, ti, dist = c.match_ratio(d1, d2, self.ratio_threshold)
        p1 = np.stack([k1["x"][qi], k1["y"][qi]], 1).astype(np.float32)
        p2 = np.stack([k2["x"][ti], k2["y"][ti]], 1).astype(np.float32)
        return MatchResult(p1, p2, qi, ti, dist)


class CameraIntrinsics:
    def __init__(self, fx=0.0, fy=0.0, cx=0.0, cy=0.0):
        self.fx, self.fy, self.cx, self.cy = float(fx), float(fy), float(cx), float(cy)

    @staticmethod
    def load_from_yaml(filepath):
        """fx, fy, cx, cy from an OpenCV-FileStorage style YAML (ref:src/motion_estimator.cpp:28-42)."""
        try:
            text = open(filepath).read()
        except OSError:
            raise RuntimeError(f"Could not open camera file: {filepath}") from None
        k = CameraIntrinsics()
        for line in text.splitlines():
            if ":" not in line or line[:1] in "%#":
                continue
            key, _, val = line.partition(":")
            try:
                v = float(val.split()[0])
            except (ValueError, IndexError):
                continue
            if key.strip() in ("fx", "fy", "cx", "cy"):
                setattr(k, key.strip(), v)
        return k

    def camera_matrix(self):
        return np.array([[self.fx, 0, self.cx], [0, self.fy, self.cy], [0, 0, 1]], np.float64)

    def _k4(self):
        return (self.fx, self.fy, self.cx, self.cy)

    def __repr__(self):
        return f"CameraIntrinsics(fx={self.fx}, fy={self.fy}, cx={self.cx}, cy={self.cy})"


class MotionEstimatorConfig:
    def __init__(self, ransac_threshold=1.0, ransac_confidence=0.999):
        self.ransac_threshold, self.ransac_confidence = float(ransac_threshold), float(ransac_confidence)

    def __repr__(self):
        return f"MotionEstimatorConfig(ransac_threshold={self.ransac_threshold}, ransac_confidence={self.ransac_confidence})"


class MotionEstimate:
    def __init__(self, rotation=None, translation=None, inliers=0, valid=False):
        self.rotation = np.zeros((3, 3)) if rotation is None else np.asarray(rotation, np.float64).reshape(3, 3)
        self.translation = np.zeros(3) if translation is None else np.asarray(translation, np.float64).reshape(3)
        self.inliers, self.valid = int(inliers), bool(valid)

    def __bool__(self):
        return self.valid


def estimate_motion(points1, points2, intrinsics, config=None):
    """Essential-matrix RANSAC + pose recovery; fewer than five points -> an invalid estimate."""
    config = config or MotionEstimatorConfig()
    p1 = np.asarray(points1, np.float32).reshape(-1, 2)
    p2 = np.asarray(points2, np.float32).reshape(-1, 2)
    if len(p1) < 5 or len(p2) < 5:
        return MotionEstimate()
    n = min(len(p1), len(p2))
    r = _ctx().estimate_motion(p1[:n], p2[:n], intrinsics._k4(), config.ransac_threshold, config.ransac_confidence)
    return MotionEstimate(r["R"], r["t"], r["inliers"], r["valid"]) if r["valid"] else MotionEstimate()


class SequenceStep:
    def __init__(self, keypoints1, keypoints2, matches, motion):
        self.keypoints1, self.keypoints2, self.matches, self.motion = keypoints1, keypoints2, matches, motion


def process_sequence(frames, intrinsics, max_features=default_max_features, ratio_threshold=default_ratio_threshold,
                     config=None):
    """Consecutive frames (n x H x W uint8) -> one SequenceStep per pair (i, i+1); one GPU call,
    every frame detected once."""
    config = config or MotionEstimatorConfig()
    frames = np.ascontiguousarray(frames, np.uint8)
    if frames.ndim != 3 or len(frames) < 2:
        raise RuntimeError("process_sequence needs at least two equally sized grayscale frames")
    rec = _ctx().pipeline_sequence(frames, int(max_features), intrinsics._k4(), float(ratio_threshold),
                                   config.ransac_threshold, config.ransac_confidence)
    return [SequenceStep(int(r["n_kp1"]), int(r["n_kp2"]), int(r["n_matches"]),
                         MotionEstimate(r["R"], r["t"], r["inliers"], True) if r["valid"] else MotionEstimate())
            for r in rec]


class Pose:
    def __init__(self, rotation=None, translation=None, timestamp=0.0):
        self.rotation = np.eye(3) if rotation is None else np.asarray(rotation, np.float64).reshape(3, 3)
        self.translation = np.zeros(3) if translation is None else np.asarray(translation, np.float64).reshape(3)
        self.timestamp = float(timestamp)

    @staticmethod
    def identity(ts=0.0):
        return Pose(timestamp=ts)

    def compose(self, motion):
        return Pose(self.rotation @ motion.rotation, self.rotation @ motion.translation + self.translation)


class Trajectory:
    """Absolute poses from relative motions; JSON layout of ref:src/trajectory.cpp:68-115."""

    def __init__(self, initial_timestamp=0.0):
        self._poses = [Pose.identity(initial_timestamp)]

    def add_motion(self, motion, timestamp=0.0):
        if not motion.valid:
            return False
        p = self._poses[-1].compose(motion)
        p.timestamp = float(timestamp)
        self._poses.append(p)
        return True

    @property
    def poses(self):
        return self._poses

    @property
    def current_pose(self):
        return self._poses[-1]

    def size(self):
        return len(self._poses)

    def __len__(self):
        return len(self._poses)

    def empty(self):
        return len(self._poses) <= 1

    def reset(self):
        self._poses = [Pose.identity()]

    def to_json(self):
        num = lambda v: "%.10g" % v
        out = ["{\n  \"poses\": [\n"]
        for i, p in enumerate(self._poses):
            out.append("    {\n      \"timestamp\": %.10f,\n      \"rotation\": [\n" % p.timestamp)
            for r in range(3):
                out.append("        [" + ", ".join(num(p.rotation[r, c]) for c in range(3)) + ("],\n" if r < 2 else "]\n"))
            out.append("      ],\n      \"translation\": [" + ", ".join(num(v) for v in p.translation) + "]\n    }")
            out.append(",\n" if i + 1 < len(self._poses) else "\n")
        out.append("  ]\n}\n")
        return "".join(out)

    def save_to_json(self, filepath):
        try:
            with open(filepath, "w") as f:
                f.write(self.to_json())
        except OSError:
            raise RuntimeError(f"Failed to open file for writing: {filepath}") from None


class TimestampedImage:
    def __init__(self, image=None, timestamp=0.0):
        self.image, self.timestamp = image, timestamp


class TimestampedImagePair:
    def __init__(self, first=None, second=None):
        self.first, self.second = first or TimestampedImage(), second or TimestampedImage()


def read_gray_image(path):
    """8-bit grayscale of a PNG/PGM/PPM/JPEG file, reduced like the C++ loader (high byte of 16-bit
    samples, BT.601 fixed-point weights for colour; of a sequential or progressive JPEG the luma plane
    as libjpeg decodes it, CMYK files refused); None if it cannot be decoded."""
    try:
        from PIL import Image
        with Image.open(path) as im:
            if im.format == "JPEG":
                if im.mode not in ("L", "RGB"):
                    return None
                im.draft("L", im.size)
                import warnings
                from PIL import ImageFile
                assert not ImageFile.LOAD_TRUNCATED_IMAGES
                with warnings.catch_warnings():
                    warnings.simplefilter("error")
                    im.load()   # a truncated file raises (refused, as in the C++ loader)
                return np.asarray(im).copy() if im.mode == "L" else None
            if im.format not in ("PNG", "PPM"):
                return None
            if im.mode in ("I;16", "I;16B", "I"):
                return (np.asarray(im).astype(np.uint32) >> 8).astype(np.uint8)
            if im.mode in ("L", "1"):
                return np.asarray(im.convert("L"))
            if im.mode == "LA":
                return np.ascontiguousarray(np.asarray(im)[..., 0])
            rgb = np.asarray(im.convert("RGB")).astype(np.int64)
            return ((rgb[..., 0] * 4899 + rgb[..., 1] * 9617 + rgb[..., 2] * 1868 + 8192) >> 14).astype(np.uint8)
    except Exception:
        return None


class image_loader:
    """Sorted .png/.jpg/.jpeg files of a directory, TUM rgb.txt timestamps from its parent
    (ref:src/image_loader.cpp:58-216)."""

    def __init__(self, directory, paths, timestamps):
        self._dir, self._paths, self._ts, self._cur = directory, paths, timestamps, 0

    @staticmethod
    def create(image_directory):
        d = str(image_directory)
        if not os.path.exists(d):
            raise RuntimeError(f"Image directory does not exist: {d}")
        parent = os.path.dirname(d.rstrip("/")) if os.path.dirname(d.rstrip("/")) else ""
        stamps = {}
        rgb_txt = os.path.join(parent, "rgb.txt")
        if os.path.exists(rgb_txt):
            for line in open(rgb_txt):
                if not line.strip() or line.startswith("#"):
                    continue
                parts = line.split()
                try:
                    stamps[os.path.join(parent, parts[1])] = float(parts[0])
                except (IndexError, ValueError):
                    pass
        paths = sorted(os.path.join(d, f) for f in os.listdir(d)
                       if os.path.isfile(os.path.join(d, f)) and os.path.splitext(f)[1].lower() in (".png", ".jpg", ".jpeg"))
        return image_loader(d, paths, [stamps.get(p, 0.0) for p in paths])

    def load_image(self, index):
        if index >= len(self._paths):
            raise RuntimeError(f"Image index out of range: {index} >= {len(self._paths)}")
        img = read_gray_image(self._paths[index])
        if img is None:
            raise RuntimeError(f"Failed to load image: {self._paths[index]}")
        return img

    def load_image_with_timestamp(self, index):
        return TimestampedImage(self.load_image(index), self.get_timestamp(index))

    def load_image_pair(self, index):
        return self.load_image(index), self.load_image(index + 1)

    def load_image_pair_with_timestamps(self, index):
        return TimestampedImagePair(self.load_image_with_timestamp(index), self.load_image_with_timestamp(index + 1))

    def next_pair(self):
        pair = self.load_image_pair(self._cur)
        self._cur += 1
        return pair

    def next_pair_with_timestamps(self):
        pair = self.load_image_pair_with_timestamps(self._cur)
        self._cur += 1
        return pair

    def has_next(self):
        return self._cur + 1 < len(self._paths)

    def reset(self):
        self._cur = 0

    def size(self):
        return len(self._paths)

    def __len__(self):
        return len(self._paths)

    def get_timestamp(self, index):
        return self._ts[index] if index < len(self._ts) else 0.0

    def has_timestamps(self):
        return any(t != 0.0 for t in self._ts)

    def __iter__(self):
        self.reset()
        return self

    def __next__(self):
        if not self.has_next():
            raise StopIteration
        return self.next_pair()
