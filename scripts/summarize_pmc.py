#!/usr/bin/env python3
"""Turns the rocprofv3 --pmc passes of scripts/pmc_passes.sh into profiles/<round>_pmc_summary.json and
profiles/traffic_<round>.json (the `roofline.traffic` input of bench.py: per ask population, the HBM bytes per launch of
every engine kernel and of the whole step).

Units / corrections (MI355X_MICROARCH.md §HBM): FETCH_SIZE and WRITE_SIZE are reported in KiB. WRITE_SIZE is calibrated
here on the fill probe (a kernel that writes exactly 1e6 x 6272 B): measured/expected is stored as `write_calibration`.
On gfx950 FETCH_SIZE reports half of the bytes of wide coalesced reads, so reads are doubled (upper bound)."""
import collections
import csv
import glob
import json
import os
import sys

src = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/pmc_r03"
rnd = sys.argv[2] if len(sys.argv) > 2 else "r03"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKLOADS = ("default", "own_template_per_ask", "unique_request_vectors", "configs4_one_gpu")
SIZES = {"configs4_one_gpu": (5_000_000, 100_000)}  # (pods, nodes); the others: 1 000 000 x 50 000
TIMER_LABEL = {"k_combine_wave": "k_combine", "k_fix_rows": "k_expand_bands",
               "k_dim_prefix_max": "k_dim_walk", "k_dim_walk_window": "k_dim_walk", "k_rank_hist": "k_rank", "k_rank_scan": "k_rank",
               "k_rank_fill": "k_rank", "k_rank_final": "k_rank", "k_decide_groups": "k_decide"}


def per_kernel(tag, counter):
    files = glob.glob(os.path.join(ROOT, src, f"{tag}_{counter}", "*", "*counter_collection.csv"))
    if not files:
        return None
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(files[0])):
        if r["Counter_Name"] == counter:
            agg[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
    return {k: {"launches": len(v), "avg_KiB": sum(v) / len(v), "min_KiB": min(v), "max_KiB": max(v)} for k, v in agg.items()}


out = {"source": src, "units": "KiB per launch (rocprofv3 FETCH_SIZE / WRITE_SIZE)"}
fill_w = per_kernel("fill", "WRITE_SIZE")
expected_kib = 1_000_000 * 784 * 8 / 1024
ratio = fill_w["fill_linear"]["avg_KiB"] / expected_kib
out["write_calibration"] = {"kernel": "fill_linear (writes exactly 6 272 000 000 B)", "reported_KiB": fill_w["fill_linear"]["avg_KiB"],
                            "expected_KiB": expected_kib, "ratio": ratio}
traffic = {"round": rnd, "pods": 1_000_000, "nodes": 50_000,
           "note": "per launch; WRITE_SIZE calibrated on a known fill; FETCH_SIZE doubled per the gfx950 wide-read correction (upper bound)",
           "workloads": {}}
for wl in WORKLOADS:
    w, f = per_kernel(wl, "WRITE_SIZE"), per_kernel(wl, "FETCH_SIZE")
    if not w or not f:
        continue
    out[wl + "_WRITE_SIZE"], out[wl + "_FETCH_SIZE"] = w, f
    kernels = {}
    for k in w:
        if "ykk::" not in k:
            continue
        short = k.split("::")[-1].split("<")[0]
        short = TIMER_LABEL.get(short, short)  # kernels the engine times under one label (bench.py looks traffic up by that label)
        wb = w[k]["avg_KiB"] * 1024 / ratio
        fb = f.get(k, {"avg_KiB": 0.0})["avg_KiB"] * 1024
        e = kernels.setdefault(short, {"write_bytes": 0.0, "fetch_bytes_raw": 0.0, "launches_per_step": 0.0, "_top": 0.0})
        steps = 4.0  # --steps 3 --warmup 1
        mine_w, mine_f = wb * w[k]["launches"] / steps, fb * f.get(k, {"launches": 0})["launches"] / steps
        e["write_bytes"] += mine_w
        e["fetch_bytes_raw"] += mine_f
        # a label's launches are those of its heaviest kernel (its helpers — descriptor / fix-up / sort kernels — add bytes only):
        # hbm_bytes / launches_per_step = the bytes of one timed region of that label
        if mine_w + 2 * mine_f >= e["_top"]:
            e["_top"] = mine_w + 2 * mine_f
            e["launches_per_step"] = w[k]["launches"] / steps
    # wave-level VALU instructions per step (its own --pmc SQ_INSTS_VALU pass, when scripts/r06_final.sh ran it): the int-ops/eval of
    # SURVEY §8(d) = valu_insts x 64 lanes / (pods x nodes)
    v = per_kernel(wl, "SQ_INSTS_VALU")
    step_valu = None
    if v:
        step_valu = 0.0
        for k in v:
            if "ykk::" not in k:
                continue
            short = TIMER_LABEL.get(k.split("::")[-1].split("<")[0], k.split("::")[-1].split("<")[0])
            mine = v[k]["avg_KiB"] * v[k]["launches"] / 4.0  # (per_kernel's field name says KiB; for this counter it is a plain count)
            step_valu += mine
            if short in kernels:
                kernels[short]["valu_insts"] = kernels[short].get("valu_insts", 0.0) + mine
    for e in kernels.values():
        e["hbm_bytes"] = int(e["write_bytes"] + 2 * e["fetch_bytes_raw"])
        e["write_bytes"], e["fetch_bytes_raw"] = int(e["write_bytes"]), int(e["fetch_bytes_raw"])
        e.pop("_top", None)
    pods, nodes = SIZES.get(wl, (1_000_000, 50_000))
    traffic["workloads"][wl] = {"pods": pods, "nodes": nodes, "kernels_per_step": kernels, "step_hbm_bytes": int(sum(e["hbm_bytes"] for e in kernels.values()))}
    if step_valu is not None:
        traffic["workloads"][wl]["step_valu_insts"] = step_valu
        traffic["workloads"][wl]["int_ops_per_eval"] = step_valu * 64 / (float(pods) * nodes)
json.dump(out, open(os.path.join(ROOT, "profiles", f"{rnd}_pmc_summary.json"), "w"), indent=1)
json.dump(traffic, open(os.path.join(ROOT, "profiles", f"traffic_{rnd}.json"), "w"), indent=1)
print(json.dumps({wl: {"step_hbm_bytes": t["step_hbm_bytes"], "top": sorted(((k, v["hbm_bytes"]) for k, v in t["kernels_per_step"].items()), key=lambda x: -x[1])[:4]}
                  for wl, t in traffic["workloads"].items()}, indent=1))
