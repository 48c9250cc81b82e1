#!/usr/bin/env python3
"""Turns the rocprofv3 --pmc passes of scripts/pmc_passes.sh into profiles/<round>_pmc_summary.json and
profiles/traffic_<round>.json (the `roofline.traffic` input of bench.py).

Units / corrections (MI355X_MICROARCH.md §HBM): FETCH_SIZE and WRITE_SIZE are reported in KiB. WRITE_SIZE is calibrated
here on the fill probe (kernels that write exactly 1e6 x 6272 B): measured/expected is stored as `write_calibration`.
On gfx950 FETCH_SIZE reports half of the bytes of wide coalesced reads, so reads are doubled (upper bound)."""
import collections
import csv
import glob
import json
import os
import sys

src = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/pmc_r01"
rnd = sys.argv[2] if len(sys.argv) > 2 else "r01"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def per_kernel(tag, counter):
    f = glob.glob(os.path.join(ROOT, src, f"{tag}_{counter}", "*", "*counter_collection.csv"))[0]
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter:
            agg[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
    return {k: {"launches": len(v), "avg_KiB": sum(v) / len(v), "min_KiB": min(v), "max_KiB": max(v)} for k, v in agg.items()}


out = {"source": src, "units": "KiB per launch (rocprofv3 FETCH_SIZE / WRITE_SIZE)"}
fill_w = per_kernel("fill", "WRITE_SIZE")
expected_kib = 1_000_000 * 784 * 8 / 1024
out["write_calibration"] = {"kernel": "fill_linear (writes exactly 6 272 000 000 B)", "reported_KiB": fill_w["fill_linear"]["avg_KiB"],
                            "expected_KiB": expected_kib, "ratio": fill_w["fill_linear"]["avg_KiB"] / expected_kib}
out["bench_WRITE_SIZE"] = per_kernel("bench", "WRITE_SIZE")
out["bench_FETCH_SIZE"] = per_kernel("bench", "FETCH_SIZE")
# the dominant kernel of the step = the one that writes the bitmap (largest WRITE_SIZE among the engine's kernels)
engine_kernels = {k: v for k, v in out["bench_WRITE_SIZE"].items() if "ykk::" in k or k.startswith("k_")}
key = max(engine_kernels, key=lambda k: engine_kernels[k]["avg_KiB"])
short = key.split("::")[-1].split("<")[0]
w = out["bench_WRITE_SIZE"][key]["avg_KiB"] * 1024 / out["write_calibration"]["ratio"]
f_raw = out["bench_FETCH_SIZE"][key]["avg_KiB"] * 1024
traffic = {"round": rnd, "kernel": short, "pods": 1_000_000, "nodes": 50_000,
           "write_bytes_per_launch": int(w), "fetch_bytes_per_launch_raw": int(f_raw),
           "fetch_bytes_per_launch_corrected_x2": int(2 * f_raw),
           "hbm_bytes_per_launch": int(w + 2 * f_raw),
           "note": "WRITE_SIZE calibrated 1.000 on a known fill; FETCH_SIZE doubled per the gfx950 wide-read correction (upper bound)"}
json.dump(out, open(os.path.join(ROOT, "profiles", f"{rnd}_pmc_summary.json"), "w"), indent=1)
json.dump(traffic, open(os.path.join(ROOT, "profiles", f"traffic_{rnd}.json"), "w"), indent=1)
print(json.dumps(traffic, indent=1))
