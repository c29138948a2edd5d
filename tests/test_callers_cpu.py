"""Callers either side of the hot path, host-only: the trajectory cases of the reference's
tests/test_trajectory.cpp and the image loader's contract (tests/test_image_loader.cpp: sorted
scan, rgb.txt timestamps, error strings), with the PNG/PGM decoders checked pixel by pixel
against Pillow.  Colour-to-gray uses the BT.601 fixed-point weights; OpenCV's imread is not
available to compare with, so that reduction is unpinned against the reference."""
import io
import os
import struct
import subprocess
import zlib

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "tests", "cpp", "test_callers")


def _fnv(arr):
    h = 1469598103934665603
    for v in arr.reshape(-1).tolist():
        h = ((h ^ v) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return h


def _gray(rgb):
    rgb = rgb.astype(np.int64)
    return ((rgb[..., 0] * 4899 + rgb[..., 1] * 9617 + rgb[..., 2] * 1868 + 8192) >> 14).astype(np.uint8)


def _adam7_png(gray):
    """8-bit gray Adam7-interlaced PNG written by hand (Pillow cannot write interlaced files)."""
    h, w = gray.shape
    raw = b""
    for x0, y0, dx, dy in [(0, 0, 8, 8), (4, 0, 8, 8), (0, 4, 4, 8), (2, 0, 4, 4), (0, 2, 2, 4), (1, 0, 2, 2), (0, 1, 1, 2)]:
        sub = gray[y0::dy, x0::dx]
        if sub.size == 0:
            continue
        prev = np.zeros(sub.shape[1], np.int64)
        for k, row in enumerate(sub):
            ft = k % 5   # exercise every filter type
            cur = row.astype(np.int64)
            left = np.concatenate([[0], cur[:-1]])
            ul = np.concatenate([[0], prev[:-1]])
            if ft == 0: pred = np.zeros_like(cur)
            elif ft == 1: pred = left
            elif ft == 2: pred = prev
            elif ft == 3: pred = (left + prev) >> 1
            else:
                p = left + prev - ul
                pa, pb, pc = abs(p - left), abs(p - prev), abs(p - ul)
                pred = np.where((pa <= pb) & (pa <= pc), left, np.where(pb <= pc, prev, ul))
            raw += bytes([ft]) + ((cur - pred) & 0xFF).astype(np.uint8).tobytes()
            prev = cur

    def chunk(t, d):
        return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d))
    ihdr = struct.pack(">IIBBBBB", w, h, 8, 0, 0, 0, 1)
    z = zlib.compress(raw)
    return b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", ihdr) + chunk(b"IDAT", z[: len(z) // 2]) + chunk(b"IDAT", z[len(z) // 2:]) + chunk(b"IEND", b"")


@pytest.fixture(scope="module")
def dataset(tmp_path_factory):
    from PIL import Image
    root = tmp_path_factory.mktemp("seq")
    d = root / "rgb"
    d.mkdir()
    rng = np.random.default_rng(11)
    expect = {}
    g8 = rng.integers(0, 256, (23, 37), dtype=np.uint8)
    Image.fromarray(g8).save(d / "00_gray8.png")
    expect["00_gray8.png"] = g8
    rgb = rng.integers(0, 256, (19, 41, 3), dtype=np.uint8)
    Image.fromarray(rgb).save(d / "01_rgb.png")
    expect["01_rgb.png"] = _gray(rgb)
    g16 = rng.integers(0, 65536, (17, 29), dtype=np.uint16)
    Image.fromarray(g16).save(d / "02_gray16.png")
    expect["02_gray16.png"] = (g16 >> 8).astype(np.uint8)
    pal = Image.fromarray(rgb).quantize(16)
    pal.save(d / "03_palette.PNG")
    expect["03_palette.PNG"] = _gray(np.asarray(pal.convert("RGB")))
    bw = rng.integers(0, 2, (13, 21), dtype=np.uint8).astype(bool)
    Image.fromarray(bw).save(d / "04_bilevel.png")
    expect["04_bilevel.png"] = (bw * 255).astype(np.uint8)
    rgba = rng.integers(0, 256, (9, 11, 4), dtype=np.uint8)
    Image.fromarray(rgba, "RGBA").save(d / "05_rgba.png")
    expect["05_rgba.png"] = _gray(rgba[..., :3])
    la = rng.integers(0, 256, (9, 11, 2), dtype=np.uint8)
    Image.fromarray(la, "LA").save(d / "06_la.png")
    expect["06_la.png"] = la[..., 0]
    inter = rng.integers(0, 256, (21, 19), dtype=np.uint8)
    (d / "07_adam7.png").write_bytes(_adam7_png(inter))
    expect["07_adam7.png"] = inter
    assert np.array_equal(np.asarray(Image.open(d / "07_adam7.png")), inter)   # the hand-made file is a valid PNG
    pgm = rng.integers(0, 256, (8, 12), dtype=np.uint8)
    (d / "08_pgm.png").write_bytes(b"P5\n# comment\n12 8\n255\n" + pgm.tobytes())
    expect["08_pgm.png"] = pgm
    # JPEG: the luma plane as libjpeg delivers it (Pillow's draft mode "L" asks libjpeg-turbo for
    # grayscale output, which is what cv::imread(IMREAD_GRAYSCALE) does)
    def jpeg_luma(path):
        im = Image.open(path)
        im.draft("L", im.size)
        im.load()
        assert im.mode == "L"
        return np.asarray(im).copy()
    yy, xx = np.mgrid[0:217, 0:333]
    photo = np.stack([(xx * 3 + yy) % 256, (xx + yy * 2) % 256, (xx * yy // 7) % 256], -1).astype(np.uint8)
    photo[40:120, 60:200] = rng.integers(0, 256, (80, 140, 3), dtype=np.uint8)
    Image.fromarray(rgb).save(d / "09_photo.jpg", quality=90)                        # 41 x 19, 4:2:0
    Image.fromarray(photo).save(d / "09a_420.jpg", quality=85)                        # odd size, 4:2:0
    Image.fromarray(photo).save(d / "09b_444.jpeg", quality=95, subsampling=0)
    Image.fromarray(photo).save(d / "09c_422.jpg", quality=60, subsampling=1)
    Image.fromarray(photo[..., 0]).save(d / "09d_gray.jpg", quality=75)
    Image.fromarray(photo).save(d / "09e_optimized.jpg", quality=80, optimize=True)   # file-specific Huffman tables
    try:
        Image.fromarray(photo).save(d / "09f_restart.jpg", quality=80, restart_marker_blocks=7)
    except TypeError:
        Image.fromarray(photo).save(d / "09f_restart.jpg", quality=80)
    for name in ("09_photo.jpg", "09a_420.jpg", "09b_444.jpeg", "09c_422.jpg", "09d_gray.jpg", "09e_optimized.jpg",
                 "09f_restart.jpg"):
        expect[name] = jpeg_luma(d / name)
    # progressive files (SOF2): spectral selection and successive approximation, DC scans interleaved,
    # AC scans per component, with and without chroma, restart markers inside the scans
    def save_progressive(img, name, **kw):   # (through a buffer: libjpeg cannot suspend a progressive pass on a small file buffer)
        buf = io.BytesIO()
        Image.fromarray(img).save(buf, "JPEG", progressive=True, **kw)
        (d / name).write_bytes(buf.getvalue())
    save_progressive(photo, "09g_progressive.jpg", quality=80)
    save_progressive(photo, "09g2_prog444.jpg", quality=88, subsampling=0)   # (under Pillow's 64 KB encoder buffer)
    save_progressive(photo[..., 1], "09g3_proggray.jpg", quality=70)
    try:
        save_progressive(photo, "09g4_progrestart.jpg", quality=85, subsampling=1, restart_marker_blocks=5)
    except TypeError:
        save_progressive(photo, "09g4_progrestart.jpg", quality=85, subsampling=1)
    for name in ("09g_progressive.jpg", "09g2_prog444.jpg", "09g3_proggray.jpg", "09g4_progrestart.jpg"):
        assert b"\xff\xc2" in (d / name).read_bytes()[:1000]
        expect[name] = jpeg_luma(d / name)
    whole = (d / "09a_420.jpg").read_bytes()
    (d / "09h_truncated.jpg").write_bytes(whole[: len(whole) // 2])
    expect["09h_truncated.jpg"] = None     # a file that ends inside the entropy-coded data is refused
    (d / "10_broken.png").write_bytes(b"\x89PNG\r\n\x1a\nnot really")
    expect["10_broken.png"] = None
    # a 65-byte file whose header promises 32768 x 32768 RGBA16 pixels (8.6 GB): refused before any allocation
    def chunk(t, body):
        return struct.pack(">I", len(body)) + t + body + struct.pack(">I", zlib.crc32(t + body))
    bomb = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", 32768, 32768, 16, 6, 0, 0, 0)) + \
        chunk(b"IDAT", zlib.compress(b"\0" * 2)) + chunk(b"IEND", b"")
    (d / "11_bomb.png").write_bytes(bomb)
    expect["11_bomb.png"] = None
    (d / "notes.txt").write_text("ignored")
    (root / "rgb.txt").write_text("# timestamp filename\n1305031102.175304 rgb/00_gray8.png\n1305031102.211214 rgb/01_rgb.png\n")
    return str(d), expect


def test_trajectory_and_loader(dataset):
    d, expect = dataset
    if not os.path.exists(BIN):   # fresh checkout: the binaries are built in-tree, not committed
        subprocess.run(["make", "-C", ROOT, "tests/cpp/test_callers"], check=True, capture_output=True, timeout=900)
    out = subprocess.run([BIN, d], capture_output=True, text=True, timeout=60)
    assert out.returncode == 0, out.stdout + out.stderr
    lines = out.stdout.splitlines()
    assert lines[-1] == "ALL PASSED"
    assert lines[0] == f"count {len(expect)} timestamps 1"
    seen = [l.split() for l in lines if l.startswith("image ")]
    assert [s[1] for s in seen] == sorted(expect)          # sorted scan, other files ignored
    for s in seen:
        want = expect[s[1]]
        if want is None:
            assert s[2] == "error" and " ".join(s[3:7]) == "Failed to load image:"
            continue
        assert (int(s[2]), int(s[3])) == want.shape, s[1]
        assert int(s[4], 16) == _fnv(want), s[1]
    stamps = {s[1]: float(s[5]) for s in seen if len(s) == 6 and s[2] != "error"}
    assert stamps["00_gray8.png"] == pytest.approx(1305031102.1753, abs=1e-3)
    assert stamps["01_rgb.png"] == pytest.approx(1305031102.2112, abs=1e-3)
    assert stamps["02_gray16.png"] == 0.0
    # the pair walk stops at the first undecodable file (the truncated JPEG, index 20)
    assert lines[-2] == "pairs 19"


def test_decoders_survive_damaged_files(dataset, tmp_path):
    """The PNG / PNM / JPEG decoders under AddressSanitizer and UBSan on the fixture files with random
    damage (flipped bytes, broken headers, truncation, spliced tails): a file may be refused, nothing
    may be read or written out of bounds."""
    d, expect = dataset
    exe = str(tmp_path / "fuzz_decoders")
    build = subprocess.run(
        ["g++", "-std=c++20", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-static-libasan",
         "-I" + os.path.join(ROOT, "include"), "-o", exe, os.path.join(ROOT, "tests", "cpp", "fuzz_decoders.cpp"),
         os.path.join(ROOT, "visual_odometry_amd", "csrc", "host", "vo_jpeg.cpp"), "-lz"],
        capture_output=True, text=True, timeout=600)
    if build.returncode != 0 and "asan" in build.stderr.lower():
        pytest.skip("no AddressSanitizer runtime for g++ here")
    assert build.returncode == 0, build.stderr
    seeds = [os.path.join(d, n) for n, want in sorted(expect.items()) if want is not None]
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:allocator_may_return_null=1:max_allocation_size_mb=4096")
    run = subprocess.run([exe, "2500", "20251111"] + seeds, capture_output=True, text=True, timeout=600, env=env)
    assert run.returncode == 0, run.stdout + run.stderr[-3000:]
    assert run.stdout.startswith(f"{len(seeds)} of {len(seeds)} seed files")


def test_python_api_host_side(dataset, tmp_path):
    """The Python module's host-side classes (names of the reference's bindings): loader contract,
    decoders equal to the C++ ones, trajectory algebra and JSON text."""
    from visual_odometry_amd import api
    d, expect = dataset
    with pytest.raises(RuntimeError, match="does not exist"):
        api.image_loader.create(d + "/nope")
    loader = api.image_loader.create(d)
    assert len(loader) == len(expect) and loader.has_timestamps()
    names = sorted(expect)
    for i, n in enumerate(names):
        if expect[n] is None:
            with pytest.raises(RuntimeError, match="Failed to load image"):
                loader.load_image(i)
        else:
            assert np.array_equal(loader.load_image(i), expect[n]), n
    with pytest.raises(RuntimeError, match="out of range"):
        loader.load_image(len(expect))
    assert loader.get_timestamp(0) == pytest.approx(1305031102.175304)
    assert sum(1 for _ in zip(range(8), loader)) == 8

    t = api.Trajectory(1305031102.175304)
    assert t.empty() and len(t) == 1
    fwd = api.MotionEstimate(np.eye(3), [0, 0, 1], 100, True)
    assert not t.add_motion(api.MotionEstimate()) and t.add_motion(fwd, 1305031102.211214)
    Ry = np.array([[0, 0, 1], [0, 1, 0], [-1, 0, 0]], float)
    assert t.add_motion(api.MotionEstimate(Ry, [0, 0, 1], 50, True), 3.0) and t.add_motion(fwd, 4.0)
    assert np.allclose(t.current_pose.translation, [1, 0, 2]) and bool(fwd) and not bool(api.MotionEstimate())
    js = t.to_json()
    assert js.startswith('{\n  "poses": [\n    {\n      "timestamp": 1305031102.1753039360,\n      "rotation": [\n        [1, 0, 0],\n')
    assert js.endswith('      "translation": [1, 0, 2]\n    }\n  ]\n}\n')
    import json
    assert len(json.loads(js)["poses"]) == 4
    t.save_to_json(tmp_path / "t.json")
    assert (tmp_path / "t.json").read_text() == js
    with pytest.raises(RuntimeError, match="Failed to open"):
        t.save_to_json("/nonexistent/dir/t.json")
    cam = tmp_path / "cam.yaml"
    cam.write_text("%YAML:1.0\n---\nfx: 517.3\nfy: 516.5\ncx: 318.6\ncy: 255.3\n")
    k = api.CameraIntrinsics.load_from_yaml(cam)
    assert (k.fx, k.fy, k.cx, k.cy) == (517.3, 516.5, 318.6, 255.3) and k.camera_matrix()[0, 2] == 318.6
    with pytest.raises(RuntimeError, match="Could not open camera file"):
        api.CameraIntrinsics.load_from_yaml(tmp_path / "none.yaml")


def test_block_range_of_the_library_equals_the_python_helper():
    from visual_odometry_amd import capi, sharding
    for n in (0, 1, 5, 64, 100, 65536):
        for parts in (1, 2, 3, 8):
            for part in range(parts):
                assert capi.block_range(n, part, parts) == sharding.block_range(n, part, parts)


REF = "/root/reference"


@pytest.mark.skipif(not os.path.isdir(REF), reason="the reference checkout is not present on this machine")
def test_reference_callers_compile_and_link_unchanged(tmp_path):
    """The reference's own callers of the hot path - its command-line tool and its benchmark
    (ref:src/main.cpp, ref:benchmarks/benchmark_matchers.cpp) - compiled as they are against the
    drop-in headers (stand-in OpenCV / Eigen / tl types, no OpenCV here) and linked against the
    drop-in library: every declaration and symbol they use must exist."""
    inc = os.path.join(ROOT, "include")
    lib = os.path.join(ROOT, "visual_odometry_amd")
    for src, exe in (("src/main.cpp", "ref_main"), ("benchmarks/benchmark_matchers.cpp", "ref_benchmark")):
        out = tmp_path / exe
        cmd = ["g++", "-std=c++20", "-O1", "-I", inc, os.path.join(REF, src), "-o", str(out), "-L", lib,
               "-lvisual_odometry_hip", "-lvo_hip", f"-Wl,-rpath,{lib}"]
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-3000:]
        # argument handling runs without a GPU: no --images -> usage and exit code 1
        r = subprocess.run([str(out)], capture_output=True, text=True, timeout=60)
        assert r.returncode == 1 and "--images" in (r.stderr + r.stdout)


def test_draw_helpers_of_the_dropin():
    # ref:tests/test_feature_detector.cpp:67-78, ref:tests/test_feature_matcher.cpp:121-135 (host code)
    exe = os.path.join(ROOT, "tests", "cpp", "test_callers")
    r = subprocess.run([exe, "--draw"], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0, r.stdout + r.stderr
