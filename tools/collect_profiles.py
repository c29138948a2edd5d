#!/usr/bin/env python3
"""Copies the evidence of the last GPU runs from gpurun_out/ into profiles/ (bench line, rocprofv3
kernel statistics, HBM traffic per launch from the FETCH_SIZE / WRITE_SIZE passes).

    python tools/collect_profiles.py r02
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]


def latest(pattern):
    return sorted(glob.glob(os.path.join(R, pattern), recursive=True), key=os.path.getmtime)[-1]


def per_kernel(fn):
    """Average counter value per launch of each kernel.  The FAST stage is the exception: a launch
    group (level 0, or levels 1-7) is up to three fast_kernel launches - <0> on the sampled tiles,
    <1> on the others, <2> over the redo list - which bench.py brackets as one, so its figure is
    the sum over all fast_kernel launches divided by the number of groups (= launches of <0>)."""
    agg = collections.defaultdict(list)
    fast_total, fast_groups = 0.0, 0
    for r in csv.DictReader(open(fn)):
        name = r["Kernel_Name"].split("(")[0].split("::")[-1]
        if name.startswith("fast_kernel"):
            fast_total += float(r["Counter_Value"])
            fast_groups += name.startswith("fast_kernel<0>")
        else:
            agg[name.split("<")[0]].append(float(r["Counter_Value"]))
    out = {k: sum(v) / len(v) for k, v in agg.items()}
    if fast_groups:
        out["fast_kernel"] = fast_total / fast_groups
    return out


d = json.load(open(os.path.join(R, "gpurun_out/bench_final.json")))
shutil.copy(os.path.join(R, "gpurun_out/bench_final.json"), os.path.join(R, f"profiles/{tag}_bench_default.json"))
stats = latest("gpurun_out/prof_default/**/*kernel_stats.csv")
shutil.copy(stats, os.path.join(R, f"profiles/{tag}_kernel_stats_default.csv"))
for w in ("c3", "c5", "c1", "sparse"):        # tools/profile_workload.sh runs of the other workloads
    try:
        shutil.copy(latest(f"gpurun_out/prof_{w}/**/*kernel_stats.csv"), os.path.join(R, f"profiles/{tag}_kernel_stats_{w}.csv"))
        shutil.copy(os.path.join(R, f"gpurun_out/prof_{w}.json"), os.path.join(R, f"profiles/{tag}_bench_{w}.json"))
    except (IndexError, OSError) as e:
        print("no profile of", w, e)
for w in ("c2", "c5", "sparse"):              # tools/serial_kernel_times.sh
    if os.path.exists(os.path.join(R, f"gpurun_out/serial_{w}.txt")):
        shutil.copy(os.path.join(R, f"gpurun_out/serial_{w}.txt"), os.path.join(R, f"profiles/{tag}_serial_kernel_times_{w}.txt"))
if os.path.exists(os.path.join(R, "gpurun_out/rehearsal_2ranks.json")) and os.path.getsize(os.path.join(R, "gpurun_out/rehearsal_2ranks.json")):
    # (gloo prints its connection notes on stdout too: keep the benchmark's line only)
    line = [l for l in open(os.path.join(R, "gpurun_out/rehearsal_2ranks.json")) if l.startswith("{")][-1]
    open(os.path.join(R, f"profiles/{tag}_rehearsal_2ranks_one_gpu.json"), "w").write(line)
if glob.glob(os.path.join(R, "gpurun_out/pmc_sq_sparse/a/**/*counter_collection.csv"), recursive=True):
    import subprocess
    js = subprocess.run([sys.executable, os.path.join(R, "tools/pmc_sq_json.py"), os.path.join(R, "gpurun_out/pmc_sq_sparse/a"),
                         os.path.join(R, "gpurun_out/pmc_sq_sparse/b"), str(d["config"]["pairs_per_step_per_gpu"])], check=True,
                        capture_output=True, text=True).stdout
    rec = json.loads(js)
    rec["workload"] = "c2_640x480_2000kp/sparse"
    rec["source"] = rec["source"].replace("--no-sequence)", "--no-sequence --scene sparse)")
    json.dump(rec, open(os.path.join(R, f"profiles/{tag}_pmc_sq_sparse.json"), "w"), indent=1)
print("bench", round(d["value"]), "pairs/s,", d["roofline"]["kernel"], d["roofline"]["avg_launch_ms"], "ms by events")
for r in list(csv.DictReader(open(stats)))[:8]:
    print("  ", r["Name"][:44].ljust(44), r["Calls"], float(r["AverageNs"]) / 1e6, "ms", r["Percentage"], "%")

fe = per_kernel(latest("gpurun_out/pmc_traffic/bench_FETCH_SIZE/**/*counter_collection.csv"))
wr = per_kernel(latest("gpurun_out/pmc_traffic/bench_WRITE_SIZE/**/*counter_collection.csv"))
cf = per_kernel(latest("gpurun_out/pmc_traffic/calib_FETCH_SIZE/**/*counter_collection.csv"))
cw = per_kernel(latest("gpurun_out/pmc_traffic/calib_WRITE_SIZE/**/*counter_collection.csv"))
# FAST: two launch groups per step (level 0, upper levels); the figure is that of an average group
groups = {"fast": ("fast_kernel", 1), "blur": ("blur_kernel", 1), "describe": ("describe_kernel", 1),
          "resize": ("resize_kernel", 7)}   # seven resize launches per step
out = {
    "workload": d["config"]["workload"], "pairs_per_step": d["config"]["pairs_per_step_per_gpu"],
    "source": "tools/pmc_traffic.sh: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over "
              "bench.py --steps 3 --warmup 1 --streams 1 --no-sequence; bytes = (2*FETCH_SIZE + WRITE_SIZE) KiB.  Calibration "
              f"(tools/calib_fetch, same passes): a 1 GiB read reports FETCH_SIZE {cf['read_dword']:.0f} KiB with dword "
              f"loads and {cf['read_dwordx4']:.0f} KiB with dwordx4 loads (half the bytes), a 256 MiB dword write "
              f"reports WRITE_SIZE {cw['write_dword']:.0f} KiB (exact).",
    "bytes_per_launch": {g: int((2 * fe[k] + wr[k]) * 1024 * m) for g, (k, m) in groups.items()},
    "kib_raw": {k: {"FETCH_SIZE": fe[k], "WRITE_SIZE": wr[k]} for k in sorted(fe) if k in wr},
}
json.dump(out, open(os.path.join(R, "profiles/pmc_traffic.json"), "w"), indent=1)
print(out["bytes_per_launch"])

# instruction-mix counters per launch (tools/pmc_sq.sh passes), when they were taken
if glob.glob(os.path.join(R, "gpurun_out/pmc_sq/a/**/*counter_collection.csv"), recursive=True):
    import subprocess
    js = subprocess.run([sys.executable, os.path.join(R, "tools/pmc_sq_json.py"), os.path.join(R, "gpurun_out/pmc_sq/a"),
                         os.path.join(R, "gpurun_out/pmc_sq/b"), str(d["config"]["pairs_per_step_per_gpu"])], check=True, capture_output=True, text=True).stdout
    open(os.path.join(R, "profiles/pmc_sq.json"), "w").write(js)
    k = json.loads(js)["kernels"]
    for name in ("fast_kernel<0>", "fast_kernel<1>", "blur_kernel", "describe_kernel", "orient_kernel", "knn2_kernel", "resize_kernel"):
        if name in k:
            print("  ", name.ljust(18), "VALU wave insts/launch %.3g" % k[name].get("SQ_INSTS_VALU", float("nan")),
                  "issue frac %.2f" % k[name].get("valu_issue_frac", float("nan")))
