#!/usr/bin/env python3
"""Per-launch instruction-mix counters of the pipeline kernels as JSON.

    python tools/pmc_sq_json.py gpurun_out/pmc_sq/a gpurun_out/pmc_sq/b > profiles/pmc_sq.json

Reads the newest rocprofv3 counter_collection.csv of each pass directory (tools/pmc_sq.sh
writes them; its bench.py command launches full batches only) and averages every counter and
the dispatch duration over all dispatches of a kernel - an average launch, as in bench.py's
roofline (resize has seven launches per step; the FAST stage two launch groups, level 0 and the
upper levels, each of fast_kernel<0> on the sampled tiles, <1> on the others with the raised
threshold and <2> over the - normally empty - redo list: the three are listed separately).  `valu_issue_frac` is the share of the GPU's vector-ALU
issue slots the kernel's wavefront instructions fill while it runs: one wave64 VALU instruction
occupies a SIMD for four cycles, 256 CUs x 4 SIMDs, clock from the agent's max (2400 MHz).
"""
import collections
import csv
import glob
import json
import os
import sys

CLOCK_GHZ = 2.4
SIMDS = 256 * 4

per = collections.defaultdict(lambda: collections.defaultdict(dict))   # kernel -> dispatch -> counter -> value
meta = {}
for d in sys.argv[1:3]:
    files = sorted(glob.glob(f"{d}/**/*counter_collection.csv", recursive=True), key=os.path.getmtime)
    if not files:
        continue
    for r in csv.DictReader(open(files[-1])):
        full = r["Kernel_Name"].split("(")[0]
        name = full.split("<")[0].split("::")[-1].split(" ")[-1]
        if name.startswith("fast_kernel") and "<" in full:   # (fast_kernel<0> / <1> / <2> do different work: kept apart)
            name += "<" + full.split("<", 1)[1]
        key = (d, r["Dispatch_Id"])
        per[name][key][r["Counter_Name"]] = float(r["Counter_Value"])
        meta[key] = (int(r["Grid_Size"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))

out = {}
for name, disp in per.items():
    if not name.split("<")[0].endswith("_kernel"):
        continue
    keys = list(disp)
    acc = collections.defaultdict(list)
    for k in keys:
        for c, v in disp[k].items():
            acc[c].append(v)
    row = {c: sum(v) / len(v) for c, v in acc.items()}
    row["grid_size_max"] = max(meta[k][0] for k in keys)
    row["launches"] = len({k[1] for k in keys if k[0] == sys.argv[1]}) or len(keys)
    row["duration_ns"] = sum(meta[k][1] for k in keys) / len(keys)
    if "SQ_INSTS_VALU" in row:
        row["valu_issue_frac"] = row["SQ_INSTS_VALU"] * 4.0 / (SIMDS * row["duration_ns"] * CLOCK_GHZ)
    out[name] = row
print(json.dumps({"workload": "c2_640x480_2000kp", "pairs_per_step": int(sys.argv[3]) if len(sys.argv) > 3 else 512, "source": "tools/pmc_sq.sh (two rocprofv3 --pmc passes over bench.py --steps 3 --warmup 1 "
                            "--no-hamming --no-cpu-baseline --no-sequence), averaged by tools/pmc_sq_json.py over all "
                            "dispatches of a kernel; durations are those under counter collection",
                  "clock_ghz": CLOCK_GHZ, "simds": SIMDS, "kernels": out}, indent=1))
