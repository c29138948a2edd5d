#!/usr/bin/env python3
"""End-to-end rate of the command-line tool on a directory of PNG frames (decode + upload +
pipeline + trajectory): tools/cli_throughput.py [frames] [batch].  The frames are a synthetic
sequence written with Pillow into a temporary directory."""
import os, subprocess, sys, tempfile, time
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from PIL import Image
from visual_odometry_amd import synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 400
batch = sys.argv[2] if len(sys.argv) > 2 else "64"
cli = os.path.join(ROOT, "visual_odometry_amd", "vo_main")
with tempfile.TemporaryDirectory() as tmp:
    d = os.path.join(tmp, "rgb")
    os.mkdir(d)
    base = synth.sequence(4242, 40)
    for i in range(n):
        Image.fromarray(base[i % len(base)] if (i // len(base)) % 2 == 0 else base[len(base) - 1 - i % len(base)]).save(
            os.path.join(d, f"{i:05d}.png"))
    cam = os.path.join(tmp, "cam.yaml")
    open(cam, "w").write("%YAML:1.0\n---\nfx: {}\nfy: {}\ncx: {}\ncy: {}\n".format(*synth.TUM_K))
    out = os.path.join(tmp, "traj.json")
    for threads in ("1", "16", "1", "16"):   # the very first run also pays for paging the libraries in
        t0 = time.perf_counter()
        run = subprocess.run([cli, "--images", d, "--camera", cam, "--output", out, "--batch", batch,
                              "--decode-threads", threads], capture_output=True, text=True, timeout=600)
        dt = time.perf_counter() - t0
        assert run.returncode == 0, run.stdout + run.stderr
        tail = [ln for ln in run.stdout.splitlines() if ln.startswith("Processed")]
        print(f"decode threads {threads}: {n - 1} pairs in {dt:.2f} s = {(n - 1) / dt:.0f} frame-pairs/s end to end "
              f"(batch {batch}, process start included); {tail[-1] if tail else ''}", flush=True)
