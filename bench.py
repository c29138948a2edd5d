#!/usr/bin/env python3
"""Benchmark of the MI355X visual-odometry front end.

One step = one pass of the fused hot path (ORB detect x2 -> Hamming kNN(2) +
ratio test -> essential-matrix RANSAC + pose) over a batch of synthetic frame
pairs that already sit in HBM.  Prints ONE JSON line (rank 0).

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Frame pairs are independent, so ranks shard them with no data-path collective
(weak scaling: every rank processes --pairs pairs per step).
"""
import argparse
import concurrent.futures
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # the reference's own CPU-runnable case: a 101-frame directory walked pair by pair with
    # orb_matcher{1000} (ref:src/main.cpp:144-184); one step = the 100 consecutive pairs
    "c1_640x480_1000kp_seq100": dict(h=480, w=640, nfeat=1000, fixed=0, pairs=100, sequence=True),
    "c2_640x480_2000kp": dict(h=480, w=640, nfeat=2000, fixed=0, pairs=512, sequence=False),
    "c3_kitti_1241x376_4000kp": dict(h=376, w=1241, nfeat=4000, fixed=0, pairs=256, sequence=False),
    "c5_1080p_10000kp_4096hyp": dict(h=1080, w=1920, nfeat=10000, fixed=4096, pairs=32, sequence=False),
}
# Ceilings (MI355X): HBM3E spec peak; the matrix cores' dense fp4 peak (the matcher multiplies
# +-1 vectors held as 4-bit floats); vector ALUs: 256 CUs x 4 SIMDs, one fp64 FMA (2 flops) per
# lane every other clock at 2.4 GHz = 78.6 TFLOP/s, one packed fp32 FMA (two lanes' worth: 4 flops)
# per lane every other clock = 157.3 TFLOP/s.
PEAK = {"hbm": (8000.0, "GB/s"), "mfma": (10000.0, "TFLOP/s"), "valu_fp64": (78.6, "TFLOP/s"),
        "valu_fp32": (157.3, "TFLOP/s")}
KERNEL_OF_GROUP = {
    "resize": "resize_kernel (chain of 7)", "fast": "fast_kernel (per launch group - level 0, levels 1-7 -: <0> on the sampled tiles, fast_raise_kernel, <1> on the other tiles, fast_check_kernel, <2> over the redo list)", "select": "select_a + harris + select_b kernels",
    "blur": "blur_kernel", "describe": "orient_kernel + describe_kernel", "knn2": "knn2_mfma_kernel",
    "ratio": "ratio_kernel", "ransac_solve": "solve_front_kernel + solve_roots_kernel + solve_models_kernel (+ second-chart pass)", "ransac_score": "score_kernel",
    "ransac_other": "normalize / init / update / finish kernels",
}
BOUND_OF_GROUP = {"resize": "hbm", "fast": "hbm", "select": "hbm", "blur": "hbm", "describe": "hbm", "knn2": "mfma",
                  "ratio": "hbm", "ransac_solve": "valu_fp64", "ransac_score": "valu_fp32", "ransac_other": "valu_fp64"}
SAMPSON_FLOPS = 45.0          # operations of one (essential matrix, match) Sampson test (packed fp32 first, fp64 when close)
FIVE_POINT_FLOPS = 3.0e4      # one five-point solve: null space, 10x20 elimination, degree-10 roots, polishing (estimate)
KNN_FLOPS = 512.0             # per (query, train row): a 256-deep dot product of +-1 vectors (the Hamming distance is (256 - dot) / 2)


def level_pixels(h, w):
    lv = []
    for l in range(8):
        s = np.float32(1.2) ** l
        lv.append(int(np.rint(np.float32(w) / np.float32(s))) * int(np.rint(np.float32(h) / np.float32(s))))
    return lv


def algorithmic_work(h, w, n_images, out, stats):
    """Algorithmic bytes / lane-operations / flops of ONE step per kernel group, counted once:
    resize reads level l-1 and writes level l; FAST reads every level pixel; the blur reads and
    writes every level pixel; selection reads every FAST candidate twice (histogram, cut), an
    81-byte Harris window per survivor of the first cut (at most 2 x max_features per image) and
    writes the keypoints; describe reads the 31x31 disc and the 37x37 blurred window of each
    keypoint and writes 28 + 32 bytes - but the windows of a level's keypoints overlap, so against
    the HBM ceiling it is charged no more than one read of the pyramid and one of the blurred
    pyramid (what has to come from memory; the window bytes themselves, served by the caches, are
    reported beside it as `describe_window_bytes_through_cache`); the matcher compares every query row with every train
    row of its pair; RANSAC tests every candidate model of the hypotheses the serial loop walks
    on every match and solves one five-point problem per hypothesis."""
    lv = level_pixels(h, w)
    kps = float(np.sum(out["n_kp1"]) + np.sum(out["n_kp2"])) * n_images / (2.0 * len(out))
    cands = stats["candidates"] / max(stats["calls"], 1)
    return {
        "resize": n_images * sum(lv[l - 1] + lv[l] for l in range(1, 8)),
        "fast": n_images * sum(lv),
        "select": cands * 8 + min(cands, 2.0 * kps) * (81 + 8) + kps * 8,
        "blur": n_images * 2 * sum(lv),
        "describe": min(kps * (31 * 31 + 37 * 37), n_images * 2.0 * sum(lv)) + kps * 60,
        "knn2": KNN_FLOPS * float(np.sum(out["n_kp1"].astype(np.float64) * out["n_kp2"])),
        "ratio": float(np.sum(out["n_kp1"])) * 16 + float(np.sum(out["n_matches"])) * 28,
        "ransac_solve": FIVE_POINT_FLOPS * float(np.sum(out["ransac_iters"])),
        "ransac_score": SAMPSON_FLOPS * float(np.sum(out["models_scored"].astype(np.float64) * out["n_matches"])),
    }


def measured_traffic(workload, pairs, group):
    """HBM bytes per launch of the group's kernel from the committed counter passes, or None when
    they were taken on another configuration."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        rec = json.load(open(path))
    except OSError:
        return None
    if rec.get("workload") != workload or rec.get("pairs_per_step") != pairs:
        return None
    return rec.get("bytes_per_launch", {}).get(group)


def _gen_pair(args):
    from visual_odometry_amd import synth
    seed, h, w, sparse = args
    if sparse:
        return synth.frame_pair(seed, height=h, width=w, sparse=True, moving_fraction=0.6, n_layers=24)
    return synth.frame_pair(seed, height=h, width=w)


def _gen_sequence(args):
    from visual_odometry_amd import synth
    seed, n, h, w = args
    return synth.sequence(seed, n, height=h, width=w)


def make_frames(cfg, rank, n_distinct, sparse):
    """Distinct synthetic frames of the workload: 2 * n_distinct for pair workloads (generated on
    the host cores in parallel, before anything touches the GPU), pairs + 1 for a sequence."""
    from visual_odometry_amd import synth
    if cfg["sequence"]:   # two sequences (the steps alternate between them), each generated serially
        seeds = [(0x5EED0000 + 1000 * rank + 500 * k, cfg["pairs"] + 1, cfg["h"], cfg["w"]) for k in range(2)]
        with concurrent.futures.ProcessPoolExecutor(2) as pool:
            a, b = list(pool.map(_gen_sequence, seeds))
        return a + b
    jobs = [(0x5EED0000 + 100000 * rank + p, cfg["h"], cfg["w"], sparse) for p in range(n_distinct)]
    workers = max(1, min(16, os.cpu_count() or 1, n_distinct))
    if workers == 1:
        pairs = [_gen_pair(j) for j in jobs]
    else:
        with concurrent.futures.ProcessPoolExecutor(workers) as pool:
            pairs = list(pool.map(_gen_pair, jobs, chunksize=max(1, n_distinct // (4 * workers))))
    return [f for ab in pairs for f in ab]


KERNELS_OF_GROUP = {"resize": ["resize_kernel"], "fast": ["fast_kernel<1>", "fast_kernel<0>"],
                    "select": ["select_a_kernel", "harris_kernel", "select_b_kernel"], "blur": ["blur_kernel"],
                    "describe": ["describe_kernel", "orient_kernel"], "knn2": ["knn2_mfma_kernel"], "ratio": ["ratio_kernel"],
                    "ransac_solve": ["solve_front_kernel", "solve_roots_wave_kernel", "solve_models_kernel"],
                    "ransac_score": ["score_kernel"], "ransac_other": ["update_kernel", "finish_kernel"]}


def measured_issue(workload, pairs, group):
    """Share of the GPU's vector-ALU issue slots the group's kernels fill while they run
    (SQ_INSTS_VALU x 4 cycles over SIMD cycles of the dispatch), from the committed counter
    passes, per kernel; None when they were taken on another configuration."""
    try:
        rec = json.load(open(os.path.join(ROOT, "profiles", "pmc_sq.json")))
    except OSError:
        return None
    if rec.get("workload") != workload or rec.get("pairs_per_step") != pairs:
        return None
    got = {k: rec["kernels"][k]["valu_issue_frac"] for k in KERNELS_OF_GROUP.get(group, [])
           if k in rec.get("kernels", {}) and "valu_issue_frac" in rec["kernels"][k]}
    return got or None


def cpu_model():
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.lower().startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def pair_roofline(h, w, work, pairs, step_s):
    """The pair-level bound: the time one step's algorithmic work would take on each of the three
    ceilings the path touches, and the largest of them against the measured step.  Bytes of the
    image stages per pair = 2 images x Sum(level pixels) x (FAST read + pyramid write + blur read +
    blur write); matcher 512 matrix-core flops per query x train row; scoring 45 flops per model x
    match."""
    P_px = sum(level_pixels(h, w))
    t_hbm = pairs * 2 * 4 * P_px / (PEAK["hbm"][0] * 1e9)
    t_match = work["knn2"] / (PEAK["mfma"][0] * 1e12)
    t_score = work["ransac_score"] / (PEAK["valu_fp64"][0] * 1e12)
    terms = {"hbm": t_hbm, "mfma": t_match, "valu_fp64": t_score}
    binding = max(terms, key=terms.get)
    return {"image_bytes_per_pair": 2 * 4 * P_px, "ms_at_hbm_peak": 1e3 * t_hbm, "ms_at_fp4_matrix_peak": 1e3 * t_match,
            "ms_at_fp64_peak": 1e3 * t_score, "binding": binding, "ms_per_step": 1e3 * step_s,
            "frac": terms[binding] / step_s}


def cpu_baseline(frames, K, cfg, budget_s=12.0):
    """The CPU oracle (scalar port of the reference path) on a bounded sample of the same frames:
    one thread, then one frame pair per thread on the box's CPU share.  Both of the reference's
    definitions are reported: the whole pair (detect x2 + match + estimate_motion, what its main
    loop does per pair) and ms per match_images call with estimate_motion outside the timed
    region (ref:benchmarks/benchmark_matchers.cpp:111-135)."""
    from oracle import oracle as O
    O.set_five_point_charts(1)   # OpenCV's single chart (the parity checks use the union over four)
    nfeat, fixed = cfg["nfeat"], cfg["fixed"]
    if cfg["sequence"]:
        pair_frames = [(frames[i], frames[i + 1]) for i in range(len(frames) - 1)]
    else:
        pair_frames = [(frames[2 * i], frames[2 * i + 1]) for i in range(len(frames) // 2)]
    n_pairs = len(pair_frames)

    def one_pair(i, times=None):
        a, b = pair_frames[i % n_pairs]
        t1 = time.perf_counter()
        ka, da = O.orb_detect(a, nfeat)
        kb, db = O.orb_detect(b, nfeat)
        qi, ti, _ = O.match(da, db)
        if times is not None:
            times.append(time.perf_counter() - t1)
        p1 = np.stack([ka["x"][qi], ka["y"][qi]], 1)
        p2 = np.stack([kb["x"][ti], kb["y"][ti]], 1)
        O.estimate_motion(p1, p2, K, fixed_iters=fixed)

    n_done, t0, t_match = 0, time.perf_counter(), []
    while time.perf_counter() - t0 < budget_s and (n_done < n_pairs or not cfg["sequence"]):
        one_pair(n_done, t_match)
        n_done += 1
    dt = time.perf_counter() - t0
    out = {
        "value": n_done / dt, "unit": "frame-pairs/s", "cores": 1, "kind": "port", "cpu_model": cpu_model(),
        "sample": f"{n_done} of the workload's {n_pairs} distinct frame pairs, detect x2 + match + E-RANSAC + pose, "
                  f"{dt:.1f} s on one host thread (oracle/vo_oracle.cpp, g++ -O3 -mavx2 -mpopcnt, no fast-math; every frame detected twice, as the "
                  f"reference's loop does)",
        "ms_per_pair": 1e3 * dt / max(n_done, 1),
        "ms_per_match_images": 1e3 * float(np.sum(t_match)) / max(n_done, 1),
    }
    threads = max(1, min(16, os.cpu_count() or 1))
    per_thread = max(1, int(round(0.5 * budget_s * out["value"])))   # about half the budget again
    t0 = time.perf_counter()
    with concurrent.futures.ThreadPoolExecutor(threads) as pool:   # the oracle is reached through ctypes: no GIL
        list(pool.map(one_pair, range(threads * per_thread)))
    dt = time.perf_counter() - t0
    out["multi_thread"] = {"value": threads * per_thread / dt, "unit": "frame-pairs/s", "cores": threads,
                           "sample": f"{threads * per_thread} frame pairs on {threads} threads, {dt:.1f} s"}
    O.set_five_point_charts(4)
    return out


class Runner:
    """One workload resident in HBM: its step function, the per-group profile and the timed region."""

    def __init__(self, torch, dist, capi, ctxs, cfg, frames, P, K, world):
        self.torch, self.dist, self.capi, self.ctxs, self.cfg, self.P, self.K, self.world = \
            torch, dist, capi, ctxs, cfg, P, K, world
        H, W = cfg["h"], cfg["w"]
        # Two image batches of different content, used alternately by consecutive steps, so that a
        # step never finds the previous step's input in the 256 MB Infinity Cache.  (`frames` holds
        # material for both when the caller generated enough: 2 (P + 1) frames of two sequences,
        # or 4 n_distinct frames of pair workloads; with less, the second batch is the first.)
        if cfg["sequence"]:
            self.n_images = P + 1
            stacks = [np.stack(frames[: P + 1])]
            if len(frames) >= 2 * (P + 1):
                stacks.append(np.stack(frames[P + 1: 2 * (P + 1)]))
        else:
            self.n_images = 2 * P
            halves = [frames] if len(frames) < 4 or (len(frames) // 2) % 2 else [frames[: len(frames) // 2], frames[len(frames) // 2:]]
            stacks = []
            for part in halves:
                n_distinct = len(part) // 2
                stacks.append(np.stack([part[2 * (p % n_distinct) + k] for p in range(P) for k in range(2)]))
        self.batches = [torch.from_numpy(st).cuda() for st in stacks]
        self.imgs = self.batches[0]
        # two record buffers per context, used alternately: a context runs the tails of two
        # consecutive calls side by side unless they write the same records
        self.ress = [torch.zeros(P * capi.PAIR_DTYPE.itemsize, dtype=torch.uint8, device="cuda")
                     for _ in range(2 * len(ctxs))]
        self.counter = 0
        self.H, self.W = H, W
        self.scene_counters = True

    def call(self, ctx, res, imgs=None):
        c = self.cfg
        fn = ctx.pipeline_sequence_dev if c["sequence"] else ctx.pipeline_pairs_dev
        n = self.P + 1 if c["sequence"] else self.P
        fn((self.imgs if imgs is None else imgs).data_ptr(), n, self.H, self.W, self.W, self.H * self.W, c["nfeat"], 0.75,
           self.K, 1.0, 0.999, c["fixed"], res.data_ptr())

    def step(self):
        k = self.counter % len(self.ctxs)
        r = self.counter % len(self.ress)
        imgs = self.batches[self.counter % len(self.batches)]
        self.counter += 1
        self.call(self.ctxs[k], self.ress[r], imgs)

    def fence(self):
        for c in self.ctxs:
            c.sync()
        self.torch.cuda.synchronize()
        if self.world > 1:
            self.dist.barrier()

    def results(self, k=0):
        return np.frombuffer(self.ress[k].cpu().numpy().tobytes(), self.capi.PAIR_DTYPE)

    def profile(self, reps=3):
        """Untimed pass: every kernel group bracketed by events, one stream, nothing overlapped.
        Returns ms per step of each group (the RANSAC chain split into solve / score / rest in
        a second pass, because the split brackets sit inside the chain's bracket) and the FAST
        stage's scene counters."""
        ctx, res, capi = self.ctxs[0], self.ress[0], self.capi
        if self.scene_counters:
            ctx.fast_stats(True)      # (the scene counters are global atomics: counted in a call of their own)
            self.call(ctx, res)
            ctx.sync()
            stats = ctx.fast_stats(False)
            stats["calls"] = 1
        else:                         # counter-collection runs: no launch that differs from the timed ones
            stats = {"pixels": 0, "pretest_pass": 0, "corners": 0, "candidates": 0, "calls": 0}
        ctx.set_timing(True)
        for _ in range(reps):
            self.call(ctx, res)
            ctx.sync()
        ms = {}
        for g, name in enumerate(capi.T_TOP):
            t, n = ctx.kernel_time_ms(g)
            ms[name] = t * n / reps if n else 0.0
        ctx.set_timing(True, groups=["ransac_solve", "ransac_score"])
        for _ in range(reps):
            self.call(ctx, res)
            ctx.sync()
        for name in ("ransac_solve", "ransac_score"):
            t, n = ctx.kernel_time_ms(capi.T_GROUPS.index(name))
            ms[name] = t * n / reps if n else 0.0
        ms["ransac_other"] = max(ms["ransac"] - ms["ransac_solve"] - ms["ransac_score"], 0.0)
        ctx.set_timing(False)
        return ms, stats

    def timed(self, steps, group):
        """Exactly `steps` steps between two fences; only `group` keeps its event brackets (a bracket
        costs ~10 us of stream time).  Returns (seconds, mean launch ms of the group, launches)."""
        torch, dist = self.torch, self.dist
        gi = self.capi.T_GROUPS.index(group)
        for c in self.ctxs:
            c.set_timing(True, groups=[group])
        self.fence()
        t0 = time.perf_counter()
        for _ in range(steps):
            self.step()
        self.fence()
        dt = time.perf_counter() - t0
        dom = [c.kernel_time_ms(gi) for c in self.ctxs]
        launches = sum(n for _, n in dom)
        avg_ms = sum(ms * n for ms, n in dom if n > 0) / max(launches, 1)
        for c in self.ctxs:
            c.set_timing(False)
        t = rank_max(torch, dist, dt)
        return t, avg_ms, launches


def rank_max(torch, dist, seconds):
    """Maximum of a host-side duration over the ranks (the value itself without a process group).
    The one-element tensor lives where the group's backend can reduce it: on the GPU for RCCL,
    on the host for the gloo rehearsal."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return float(seconds)
    t = torch.tensor([seconds], dtype=torch.float64, device="cpu" if dist.get_backend() == "gloo" else "cuda")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def roofline_of(group, work_per_step, avg_ms, launches, steps, workload, P):
    bound = BOUND_OF_GROUP[group]
    peak, unit = PEAK[bound]
    per_step = launches / max(steps, 1)
    per_launch = work_per_step[group] / max(per_step, 1e-9)
    scale = 1e9 if bound == "hbm" else 1e12
    achieved = per_launch / max(avg_ms * 1e-3, 1e-12) / scale
    what = {"hbm": "algorithmic_bytes_per_launch", "mfma": "algorithmic_flops_per_launch",
            "valu_fp64": "algorithmic_flops_per_launch", "valu_fp32": "algorithmic_flops_per_launch"}[bound]
    return {"bound": bound, "kernel": KERNEL_OF_GROUP[group], "achieved": achieved, "peak": peak, "unit": unit,
            "frac": achieved / peak, "traffic": measured_traffic(workload, P, group) if bound == "hbm" else None,
            "valu_issue_frac": measured_issue(workload, P, group),
            "avg_launch_ms": avg_ms, "launches_timed": launches, "launches_per_step": per_step, what: per_launch}


def scene_stats(stats, out):
    px = max(stats["pixels"], 1)
    m = np.maximum(out["n_matches"], 1)
    return {"pretest_pass_frac": stats["pretest_pass"] / px, "corner_frac": stats["corners"] / px,
            "corners_per_Mpx": 1e6 * stats["corners"] / px, "candidates_after_nms_per_Mpx": 1e6 * stats["candidates"] / px,
            "mean_keypoints": float(np.mean(out["n_kp1"])), "mean_matches": float(np.mean(out["n_matches"])),
            "mean_inliers": float(np.mean(out["inliers"])), "mean_inlier_ratio": float(np.mean(out["inliers"] / m)),
            "valid_fraction": float(np.mean(out["valid"])), "mean_ransac_iters": float(np.mean(out["ransac_iters"])),
            "mean_models_per_hypothesis": float(np.sum(out["models_scored"]) / max(np.sum(out["ransac_iters"]), 1)),
            "truncated_pairs": int(np.count_nonzero(out["status"]))}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="c2_640x480_2000kp", choices=list(WORKLOADS))
    ap.add_argument("--pairs", type=int, default=0, help="frame pairs per step per GPU")
    ap.add_argument("--distinct", type=int, default=0,
                    help="distinct synthetic pairs, tiled up to --pairs (default: all pairs distinct)")
    ap.add_argument("--scene", default="dense", choices=["dense", "sparse"],
                    help="dense: every level over-fills its keypoint quota many times, ~70 %% inliers; sparse: a few "
                         "corners per thousand pixels, layers that move on their own (~40 %% inliers)")
    ap.add_argument("--streams", type=int, default=1,
                    help="contexts (streams + workspace) the steps rotate over; one context already "
                         "overlaps a batch's RANSAC rounds (side stream) with the next batch's detection")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the step-latency distribution, the batch-size sweep and the host-buffer (PCIe-inclusive) rate")
    ap.add_argument("--no-hamming", action="store_true")
    ap.add_argument("--no-sequence", action="store_true")
    ap.add_argument("--no-aux-scene", action="store_true")
    ap.add_argument("--no-dropin", action="store_true")
    ap.add_argument("--no-configs", action="store_true",
                    help="skip the short legs on the configurations other than --workload")
    ap.add_argument("--no-scene-counters", action="store_true",
                    help="skip the one call that counts pretest passes / corners with global atomics (its FAST "
                         "launches are several times slower than normal ones and would skew per-kernel averages "
                         "of a profiler run)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")

    cfg = WORKLOADS[args.workload]
    H, W, nfeat, fixed = cfg["h"], cfg["w"], cfg["nfeat"], cfg["fixed"]
    P = cfg["pairs"] if cfg["sequence"] else (args.pairs or cfg["pairs"])
    n_distinct = P if cfg["sequence"] else max(1, min(args.distinct or P, P))
    # synthetic frames first (worker processes), the GPU runtime afterwards
    # (twice the distinct pairs of a step: consecutive steps alternate between two batches)
    frames = make_frames(cfg, rank, 2 * n_distinct, args.scene == "sparse")
    aux_frames = None
    if not args.no_aux_scene and not cfg["sequence"] and args.scene == "dense":
        aux_frames = make_frames(cfg, rank, min(64, P), True)
    # the other configurations, measured in short legs after the headline one
    other = {}
    if not args.no_configs and args.scene == "dense":
        for name, ocfg in WORKLOADS.items():
            if name != args.workload:
                other[name] = make_frames(ocfg, rank, 2 * ocfg["pairs"], False)

    import torch   # before the HIP library: one HIP runtime for both
    import torch.distributed as dist
    from visual_odometry_amd import capi, sharding, synth

    # Rehearsal on a box with fewer GPUs than ranks (VO_BENCH_SHARE_GPUS=1): ranks share the GPUs
    # there are and the process group is gloo, because RCCL refuses two ranks on one device.
    # It exercises every line of the multi-rank path except RCCL itself; its figures are not
    # scaling measurements (the ranks compete for one GPU) and the line says so.
    rehearsal = world > 1 and os.environ.get("VO_BENCH_SHARE_GPUS") == "1"
    if rehearsal:
        local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    if world > 1:
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    K = synth.default_K(W, H)
    n_streams = max(1, args.streams)
    ctxs = [capi.Context(local_rank) for _ in range(n_streams)]
    ctx = ctxs[0]
    run = Runner(torch, dist, capi, ctxs, cfg, frames, P, K, world)
    run.scene_counters = not args.no_scene_counters

    for _ in range(max(args.warmup, n_streams)):
        run.step()
    run.fence()
    kernel_ms, stats = run.profile()
    out = run.results()
    work = algorithmic_work(H, W, run.n_images, out, stats)
    dominant = max(work, key=lambda g: kernel_ms[g])
    dt, dom_avg_ms, dom_launches = run.timed(args.steps, dominant)
    out = run.results()
    value = P * args.steps * world / dt

    line = {
        "metric": "frame-pairs/sec (ORB+match+E-RANSAC, 640x480, 2k kps) + Hamming match GB/s",
        "value": value, "unit": "frame-pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "u8", "data": "synthetic",
        "config": {"workload": args.workload, "pairs_per_step_per_gpu": P, "distinct_pairs": 2 * n_distinct,
                   "image_batches_rotated": len(run.batches),
                   "scene": args.scene, "mode": "consecutive frames of a sequence, each detected once"
                   if cfg["sequence"] else "independent pairs, both images of every pair detected",
                   "image": f"{W}x{H}", "max_features": nfeat, "ratio": 0.75,
                   "ransac": "adaptive<=1000" if fixed == 0 else f"fixed {fixed}", "parallelism": f"pairs x{world}",
                   "streams_per_gpu": n_streams, "record_buffers_per_context": 2},
        "roofline": roofline_of(dominant, work, dom_avg_ms, dom_launches, args.steps, args.workload, P),
        "scene": scene_stats(stats, out),
        "pipeline": {"kernel_group_ms_per_step_single_stream": kernel_ms,
                     "algorithmic_work_per_step": work,
                     "describe_window_bytes_through_cache": float(np.sum(out["n_kp1"]) + np.sum(out["n_kp2"]))
                     * run.n_images / (2.0 * len(out)) * (31 * 31 + 37 * 37),
                     "fraction_of_group_ceiling_single_stream": {
                         g: work[g] / (kernel_ms[g] * 1e-3) / (1e9 if BOUND_OF_GROUP[g] == "hbm" else 1e12) /
                         PEAK[BOUND_OF_GROUP[g]][0] for g in work if kernel_ms[g] > 0}},
    }
    line["roofline"]["note"] = (
        "the kernel group with the largest time per step among all groups, against that group's own ceiling: HBM "
        "bytes for the image kernels, fp4 matrix-core flops (512 per query x train row: the Hamming distance as a "
        "256-deep dot product of +-1 vectors) for the matcher, "
        "fp64 vector flops (~3e4 per five-point solve) for the five-point kernels, packed fp32 vector flops (45 per "
        "model x match Sampson test, decided in single precision with a rounding bound, in fp64 when close) for the "
        "scoring kernel; "
        "achieved = algorithmic work of an average launch / mean launch duration by HIP events on the kernel's "
        "stream inside the timed steps (for the FAST stage a launch is one launch group of up to five kernels, "
        "bracketed as one: rocprofv3's durations of fast_kernel<0>, <1>, <2>, fast_raise_kernel and "
        "fast_check_kernel per step add up to launches_per_step x avg_launch_ms); traffic = 2*FETCH_SIZE + WRITE_SIZE per launch from profiles/ when the "
        "configuration matches")

    ft = ctx.fast_thresholds(0)   # of the last timed step's first image
    line["scene"]["fast_thresholds_image0"] = None if ft is None else {
        "per_level": ft[0].tolist(), "levels_run_again": int(ft[1].sum()),
        "note": "FAST threshold the unsampled tiles of a level ran with (20 = plain): raised towards the score cut "
                "of the selection, results unchanged"}
    line["pair_roofline"] = pair_roofline(H, W, work, P, dt / args.steps)

    if not args.no_extras:
        # The headline workload with the matcher on the vector ALUs only (VO_OPT_KNN_MODE = 1, the
        # xor / popcount kernel): what the step costs without matrix instructions.
        for c in ctxs:
            c.set_option("knn_mode", 1)
        for _ in range(2):
            run.step()
        run.fence()
        v_steps = max(5, args.steps // 2)
        t0 = time.perf_counter()
        for _ in range(v_steps):
            run.step()
        run.fence()
        v_dt = rank_max(torch, dist, time.perf_counter() - t0)
        for c in ctxs:
            c.set_option("knn_mode", -1)
        line["vector_alu_matcher"] = {"value": P * v_steps * world / v_dt, "unit": "frame-pairs/s",
                                      "ms_per_step": 1e3 * v_dt / v_steps, "steps": v_steps,
                                      "note": "same workload, Hamming distances by xor + popcount on the vector ALUs "
                                              "instead of fp4 matrix instructions; identical results"}
        # Synchronous steps (a fence after each): the distribution of one batch's latency.  Larger
        # than ms_per_step, where a batch's RANSAC tail runs beside the next batch's detection.
        lat = []
        for _ in range(min(max(args.steps, 5), 20)):
            t0 = time.perf_counter()
            run.step()
            run.fence()
            lat.append(1e3 * (time.perf_counter() - t0))
        line["sync_step_ms"] = {"median": float(np.median(lat)), "p10": float(np.percentile(lat, 10)),
                                "p90": float(np.percentile(lat, 90)), "steps": len(lat)}
        if not cfg["sequence"]:
            # Host buffers in, results out (rank-local, not overlapped): the image batch is copied from
            # pinned host memory, the pipeline runs, the result records are copied back.  A caller
            # that double-buffers hides the smaller of copy and compute; this is the plain sum.
            host_imgs = run.imgs.cpu().pin_memory()
            host_res = torch.empty(run.ress[0].shape, dtype=torch.uint8).pin_memory()
            reps = max(3, args.steps // 4)
            cp = []
            for it in range(reps + 1):
                if it == 1:
                    run.fence()
                    t0 = time.perf_counter()
                tc = time.perf_counter()
                run.imgs.copy_(host_imgs, non_blocking=True)
                torch.cuda.synchronize()
                cp.append(time.perf_counter() - tc)
                run.call(ctx, run.ress[0])
                ctx.sync()
                host_res.copy_(run.ress[0], non_blocking=True)
                torch.cuda.synchronize()
            hd = (time.perf_counter() - t0) / reps
            t = rank_max(torch, dist, hd)
            line["host_buffers_in_and_out"] = {
                "value": P * world / t, "unit": "frame-pairs/s", "ms_per_step": 1e3 * t,
                "h2d_ms": 1e3 * float(np.mean(cp[1:])), "h2d_GBps": host_imgs.numel() / float(np.mean(cp[1:])) / 1e9,
                "note": "copy of the image batch from pinned host memory + pipeline + copy of the results, one after "
                        "the other; never `value`"}
            # The same with two image buffers and two record buffers: the copy of batch k + 1 runs on a
            # stream of its own while batch k is in the pipeline, and the per-call waits
            # (vo_pipeline_wait) let batch k - 1's RANSAC rounds run on beside batch k's detection.
            # (One context: a second one brings the process to more streams than it has hardware
            # queues, and the copy then queues in front of kernels.)
            bufs = [run.imgs, torch.empty_like(run.imgs)]
            ress = [run.ress[0], torch.empty_like(run.ress[0])]
            copy_stream = torch.cuda.current_stream()   # (idle otherwise: a fifth stream would share a hardware queue)
            n_db = 2 * reps + 2

            main_stream = torch.cuda.ExternalStream(ctx.stream)   # the context's stream, to order it behind an upload
            copied = [torch.cuda.Event(), torch.cuda.Event()]

            def upload(k):                                  # asynchronous: the host does not wait for the copy
                with torch.cuda.stream(copy_stream):
                    bufs[k % 2].copy_(host_imgs, non_blocking=True)
                    copied[k % 2].record(copy_stream)

            def launch(k):
                main_stream.wait_event(copied[k % 2])       # the detection of batch k starts behind its upload
                ctx.pipeline_pairs_dev(bufs[k % 2].data_ptr(), P, H, W, W, H * W, nfeat, 0.75, K, 1.0, 0.999, fixed,
                                       ress[k % 2].data_ptr())
                return ctx.pipeline_ticket()

            upload(0)
            prev = launch(0)
            upload(1)
            ctx.sync()
            torch.cuda.synchronize()
            upload(0)
            prev = launch(0)
            upload(1)
            t0 = time.perf_counter()
            for k in range(1, n_db + 1):
                cur = launch(k)                             # batch k from the buffer filled during batch k - 1
                ctx.pipeline_wait(prev, inputs=True)        # batch k - 1 has read its images:
                upload(k + 1)                               # ... its buffer takes batch k + 1
                ctx.pipeline_wait(prev, results=True)       # batch k - 1 is complete: its records go back
                host_res.copy_(ress[(k - 1) % 2], non_blocking=True)
                prev = cur
            ctx.sync()
            host_res.copy_(ress[n_db % 2], non_blocking=True)
            torch.cuda.synchronize()
            hd2 = (time.perf_counter() - t0) / n_db
            t = rank_max(torch, dist, hd2)
            line["host_buffers_in_and_out"]["double_buffered"] = {
                "value": P * world / t, "unit": "frame-pairs/s", "ms_per_step": 1e3 * t,
                "note": "two image buffers and two record buffers; the next batch's upload runs beside the pipeline on "
                        "another stream (the context's stream waits for the copy's event, the host does not); per-call "
                        "waits (vo_pipeline_wait) instead of vo_sync"}
            del ress
            del bufs, host_imgs, host_res
            # Batch sizes below the headline's: a batch has to fill 256 CUs and amortise a chain of
            # some sixty launches.
            sweep = []
            for b in (1, 8, 64, 256):
                if b >= P:
                    continue
                rb = Runner(torch, dist, capi, [ctx], cfg, frames, b, K, world)
                for _ in range(3):
                    rb.step()
                rb.fence()
                nb = max(args.steps, 40 if b <= 8 else 20)
                t0 = time.perf_counter()
                for _ in range(nb):
                    rb.step()
                rb.fence()
                bd = (time.perf_counter() - t0) / nb
                t = rank_max(torch, dist, bd)
                sweep.append({"pairs_per_step_per_gpu": b, "value": b * world / t,
                              "ms_per_step": 1e3 * t})
                del rb
            line["batch_sweep"] = sweep
            # One pair per call with the host waiting for every result (a tracking loop that needs
            # the pose before it takes the next frame): the same buffers every call, the ordinary
            # launch chain against the chain replayed as one HIP graph (VO_OPT_PIPELINE_GRAPH).
            one = Runner(torch, dist, capi, [ctx], cfg, frames, 1, K, world)
            fenced = {}
            for graph in (0, 1):
                ctx.set_option("pipeline_graph", graph)
                for _ in range(5):
                    one.call(ctx, one.ress[0])
                    ctx.sync()
                n_f = 200
                t0 = time.perf_counter()
                for _ in range(n_f):
                    one.call(ctx, one.ress[0])
                    ctx.sync()
                fenced["graph_replay" if graph else "launch_chain"] = 1e3 * (time.perf_counter() - t0) / n_f
            ctx.set_option("pipeline_graph", 0)
            line["one_pair_fenced_call_ms"] = fenced
            del one

    if aux_frames is not None:
        # The other end of the scene range: few corners, ~40 % inliers (many RANSAC rounds).  Short
        # run, same code path; not the headline.
        aux = Runner(torch, dist, capi, ctxs, cfg, aux_frames, P, K, world)
        for _ in range(2):
            aux.step()
        aux.fence()
        a_ms, a_stats = aux.profile()
        a_out = aux.results()
        a_work = algorithmic_work(H, W, aux.n_images, a_out, a_stats)
        a_dom = max(a_work, key=lambda g: a_ms[g])
        a_steps = max(5, args.steps // 2)   # (a short leg over-weights the first step, which has no earlier tail beside it)
        a_dt, a_avg, a_n = aux.timed(a_steps, a_dom)
        line["aux_sparse_scene"] = {
            "value": P * a_steps * world / a_dt, "unit": "frame-pairs/s", "ms_per_step": 1e3 * a_dt / a_steps,
            "steps": a_steps, "distinct_pairs": len(aux_frames) // 2,
            "roofline": roofline_of(a_dom, a_work, a_avg, a_n, a_steps, args.workload + "/sparse", P),
            "fast_kernel_fraction_of_hbm_single_stream": a_work["fast"] / (a_ms["fast"] * 1e-3) / 1e9 / PEAK["hbm"][0],
            "scene": scene_stats(a_stats, a_out), "kernel_group_ms_per_step_single_stream": a_ms}
        del aux

    if not args.no_sequence and not cfg["sequence"]:
        # the same pipeline fed as a video sequence: 2P consecutive frames -> 2P - 1 pairs, each
        # frame detected once (what a streaming caller of the reference's main loop would use).
        # 2P frames, so that the detection launches have the size of the headline steps'.
        # (the two frames of the first pair alternate, so that every consecutive pair shows the
        # same scene under a small motion)
        F = 2 * P
        seq = torch.from_numpy(np.stack([frames[i & 1] for i in range(F)])).cuda()
        res_seq = [torch.zeros((F - 1) * capi.PAIR_DTYPE.itemsize, dtype=torch.uint8, device="cuda") for _ in range(2)]
        reps = max(3, args.steps // 4)
        for it in range(2):
            ctx.pipeline_sequence_dev(seq.data_ptr(), F, H, W, W, H * W, nfeat, 0.75, K, 1.0, 0.999, fixed,
                                      res_seq[it % 2].data_ptr())
        run.fence()
        t0 = time.perf_counter()
        for it in range(reps):
            ctx.pipeline_sequence_dev(seq.data_ptr(), F, H, W, W, H * W, nfeat, 0.75, K, 1.0, 0.999, fixed,
                                      res_seq[it % 2].data_ptr())
        run.fence()
        sd = (time.perf_counter() - t0) / reps
        t = rank_max(torch, dist, sd)
        line["sequence_mode"] = {"frames_per_step_per_gpu": F, "pairs_per_step_per_gpu": F - 1,
                                 "value": (F - 1) * world / t,
                                 "unit": "frame-pairs/s", "ms_per_step": 1e3 * t,
                                 "note": "consecutive frames, every frame detected once (vo_pipeline_sequence_dev); "
                                         "not the headline metric, which detects both images of every pair"}
        del seq, res_seq

    if other:
        # Every other configuration of the benchmark in a short leg of its own: warm-up, the serial
        # profile pass that picks the dominant kernel group, five timed steps bracketed like the
        # headline's.  `value` above stays the headline configuration's.
        line["configs"] = {}
        for name, ofr in other.items():
            ocfg = WORKLOADS[name]
            oK = synth.default_K(ocfg["w"], ocfg["h"])
            leg = Runner(torch, dist, capi, ctxs, ocfg, ofr, ocfg["pairs"], oK, world)
            for _ in range(2):
                leg.step()
            leg.fence()
            o_ms, o_stats = leg.profile(reps=2)
            o_out = leg.results()
            o_work = algorithmic_work(ocfg["h"], ocfg["w"], leg.n_images, o_out, o_stats)
            o_dom = max(o_work, key=lambda g: o_ms[g])
            o_steps = 20 if ocfg["sequence"] else 10   # (a step of the 100-pair sequence is well under a millisecond)
            o_dt, o_avg, o_n = leg.timed(o_steps, o_dom)
            o_out = leg.results()
            line["configs"][name] = {
                "value": ocfg["pairs"] * o_steps * world / o_dt, "unit": "frame-pairs/s",
                "ms_per_step": 1e3 * o_dt / o_steps, "steps": o_steps, "pairs_per_step_per_gpu": ocfg["pairs"],
                "image": f"{ocfg['w']}x{ocfg['h']}", "max_features": ocfg["nfeat"],
                "ransac": "adaptive<=1000" if ocfg["fixed"] == 0 else f"fixed {ocfg['fixed']}",
                "mode": "consecutive frames of a sequence, each detected once" if ocfg["sequence"]
                else "independent pairs, both images of every pair detected",
                "dominant_group": o_dom,
                "roofline": roofline_of(o_dom, o_work, o_avg, o_n, o_steps, name, ocfg["pairs"]),
                "kernel_group_ms_per_step_single_stream": o_ms,
                "mean_matches": float(np.mean(o_out["n_matches"])), "mean_inliers": float(np.mean(o_out["inliers"])),
                "valid_fraction": float(np.mean(o_out["valid"])), "mean_ransac_iters": float(np.mean(o_out["ransac_iters"]))}
            del leg
        other.clear()

    if not args.no_hamming:
        # second half of the metric: brute-force Hamming kNN(2), 65 536 x 65 536 descriptors.  Query
        # rows are sharded over the ranks in contiguous blocks (sharding.block_range), the train set
        # is replicated, and every repetition ends with the path's one collective: the ranks'
        # (idx2 | dist2) rows gathered on rank 0 (sharding.sharded_knn2, the function the
        # world_size-2 CPU test drives).
        nq_total, nt = 65536, 65536
        b, e = sharding.block_range(nq_total, rank, world)
        q_all = torch.from_numpy(synth.descriptors(7, nq_total)).cuda()
        tr = torch.from_numpy(synth.descriptors(8, nt)).cuda()
        idx = torch.empty((e - b, 2), dtype=torch.int32, device="cuda")
        dst = torch.empty((e - b, 2), dtype=torch.int32, device="cuda")

        # The search is enqueued on the context's stream and the gather right behind it on the same
        # stream (torch's collectives order themselves after the current stream): no host wait in
        # between; a repetition's device time is taken with events on that stream.
        main_stream = torch.cuda.ExternalStream(ctx.stream)

        def gpu_knn2(q_rows, t_rows):
            ctx.hamming_knn2_dev(q_rows.data_ptr(), q_rows.shape[0], t_rows.data_ptr(), t_rows.shape[0],
                                 idx.data_ptr(), dst.data_ptr())
            return idx, dst

        with torch.cuda.stream(main_stream):
            for _ in range(2):
                table = sharding.sharded_knn2(gpu_knn2, q_all, tr)
            run.fence()
            reps = 5
            ctx.set_timing(True, groups=["knn2"])
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0 = time.perf_counter()
            ev0.record(main_stream)
            for _ in range(reps):
                table = sharding.sharded_knn2(gpu_knn2, q_all, tr)   # kernel + (world > 1) the gather, every repetition
            ev1.record(main_stream)
            run.fence()
        wall = (time.perf_counter() - t0) / reps
        hd = rank_max(torch, dist, ev0.elapsed_time(ev1) * 1e-3 / reps)
        wall = rank_max(torch, dist, wall)
        k_ms = ctx.kernel_time_ms(capi.T_GROUPS.index("knn2"))[0]
        ctx.set_timing(False)
        # what one of eight ranks runs: an 8 192-row block of the queries against the whole train set
        # (too few query blocks for 256 compute units, so the library splits the train set as well)
        shard_ms = None
        if world == 1:
            ctx.set_timing(True, groups=["knn2"])
            for _ in range(5):
                ctx.hamming_knn2_dev(q_all.data_ptr(), 8192, tr.data_ptr(), nt, idx.data_ptr(), dst.data_ptr())
            ctx.sync()
            shard_ms = ctx.kernel_time_ms(capi.T_GROUPS.index("knn2"))[0]
            ctx.set_timing(False)
        # the same search by the vector-ALU kernel (xor + popcount on 32-byte rows, no matrix
        # instructions: VO_OPT_KNN_MODE = 1), for comparison
        valu_ms = None
        if world == 1:
            ctx.set_option("knn_mode", 1)
            ctx.set_timing(True, groups=["knn2"])
            for _ in range(3):
                ctx.hamming_knn2_dev(q_all.data_ptr(), nq_total, tr.data_ptr(), nt, idx.data_ptr(), dst.data_ptr())
            ctx.sync()
            valu_ms = ctx.kernel_time_ms(capi.T_GROUPS.index("knn2"))[0]
            ctx.set_timing(False)
            ctx.set_option("knn_mode", -1)
        alg_bytes = float(nq_total) * nt * 32
        flops = KNN_FLOPS * nq_total * nt
        line["hamming"] = {"workload": "c4_hamming_65536x65536", "value": alg_bytes / hd / 1e9,
                           "unit": "GB/s (Nq*Nt*32 B / s)", "pair_compares_per_s": float(nq_total) * nt / hd,
                           "ms": 1e3 * hd, "host_wall_ms": 1e3 * wall, "kernel_ms_rank0": k_ms, "rows_this_rank": e - b,
                           "shard_of_8192_rows_kernel_ms": shard_ms,
                           "table_rows_on_rank0": int(table.shape[0]) if table is not None else None,
                           "roofline": {"bound": "mfma", "kernel": "knn2_mfma_kernel",
                                        "achieved": flops * (e - b) / nq_total / (k_ms * 1e-3) / 1e12 * world,
                                        "peak": PEAK["mfma"][0] * world, "unit": "TFLOP/s",
                                        "frac": flops * (e - b) / nq_total / (k_ms * 1e-3) / 1e12 / PEAK["mfma"][0],
                                        "traffic": None,
                                        "note": "512 flops per query x train row (a 256-deep +-1 dot product in fp4 "
                                                "v_mfma_scale_f32_32x32x64_f8f6f4 tiles, measured at 8.3 PFLOP/s in a "
                                                "bare stream of the instruction) against the dense fp4 peak; what "
                                                "binds the kernel is the vector ALU issue beside the matrix "
                                                "instructions: two operations per distance keep the two best, seven "
                                                "per train dword turn bits into fp4 signs, and each matrix "
                                                "instruction holds the issue port for about 18 cycles"},
                           "vector_alu_kernel": None if valu_ms is None else {
                               "kernel": "knn2_bcast_kernel", "ms": valu_ms, "value": alg_bytes / (valu_ms * 1e-3) / 1e9,
                               "unit": "GB/s (Nq*Nt*32 B / s)",
                               "frac_of_int_lane_peak": 20.0 * nq_total * nt / (valu_ms * 1e-3) / 78.6e12,
                               "note": "VO_OPT_KNN_MODE = 1: 8 xor + 8 v_bcnt + top-2 per query x train row, 20 "
                                       "lane-operations against 256 CUs x 4 SIMDs x 32 lanes per clock; its "
                                       "half-rate instructions cap the mix near 0.55"},
                           "compulsory_hbm_bytes": (nq_total + nt) * 32 + nq_total * 16}
        del q_all, tr

    if rank == 0 and world == 1 and not args.no_dropin:
        # The reference benchmark's own metric on the GPU path: wall-clock ms per match_images call
        # (host images in, matched points out, estimate_motion excluded) through the fused
        # host-buffer entry point the C++ drop-in's match_images_orb uses.
        if hasattr(ctx, "match_images"):
            if cfg["sequence"]:
                pf = [(frames[i], frames[i + 1]) for i in range(min(20, len(frames) - 1))]
            else:
                pf = [(frames[2 * i], frames[2 * i + 1]) for i in range(min(20, len(frames) // 2))]
            for a, b_ in pf[:3]:
                ctx.match_images(a, b_, nfeat, 0.75)
            t0 = time.perf_counter()
            for a, b_ in pf:
                ctx.match_images(a, b_, nfeat, 0.75)
            line["dropin_ms_per_match_images"] = 1e3 * (time.perf_counter() - t0) / len(pf)

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        line["cpu_baseline"] = cpu_baseline(frames, K, cfg)
    if rehearsal:
        line["rehearsal"] = ("ranks share %d GPU(s), gloo process group: a run of the multi-rank code path, not a "
                             "scaling measurement" % torch.cuda.device_count())
    if rank == 0:
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
