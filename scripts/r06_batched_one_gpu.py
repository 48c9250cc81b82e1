"""Round 6: allocation rounds on ONE GPU in batches (YKPRED_TUNE round_batched=1: parallel proposals, the host replays the loop) against
the sequential kernel — the same decisions, the time of a 20 000-ask round of configs[2] (and of the configs[4] ask mix with --configs4,
of the reference's perf shape with --perf). Usage on the GPU box: python scripts/r06_batched_one_gpu.py [--configs4] [--perf] [asks]"""
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench  # noqa: E402

pkg = importlib.import_module("yunikorn-k8shim_amd")
n_asks = int([a for a in sys.argv[1:] if a.isdigit()][0]) if any(a.isdigit() for a in sys.argv[1:]) else 20_000
shapes = [("configs2", dict(seed=bench.SEED + 2, num_nodes=50_000, num_pods=1_000_000, num_templates=2000, node_affinity=1))]
if "--configs4" in sys.argv:
    shapes.append(("configs4_mix", dict(seed=bench.SEED + 4, num_nodes=100_000, num_pods=1_000_000, num_templates=2000, node_affinity=1, spread=1)))
out = {}
for name, kw in shapes:
    got = {}
    for knob in (0, 1):
        os.environ["YKPRED_TUNE"] = f"round_batched={knob},round_prof=1"
        pm = pkg.GpuPredicateManager(device=0)
        try:
            pm.generate_kwok(**kw)
            pm.evaluate(decisions=True)
            pm.synchronize()
            asks = np.arange(n_asks, dtype=np.int32)
            pm.allocate_round(asks=asks[:64], apply=False)
            times = []
            for _ in range(3):
                t0 = time.perf_counter()
                g = pm.allocate_round(asks=asks, apply=False)
                times.append(time.perf_counter() - t0)
            got[knob] = g
            out[f"{name}_{'batched' if knob else 'sequential'}"] = {"asks": n_asks, "round_ms": round(min(times) * 1e3, 2), "allocations_per_sec": round(n_asks / min(times)),
                                                                    "allocated": int((g >= 0).sum()), "distinct_nodes": int(len(np.unique(g[g >= 0])))}
        finally:
            pm.close()
    diff = np.flatnonzero(got[0] != got[1])
    out[f"{name}_equal"] = bool(len(diff) == 0)
    if len(diff):
        out[f"{name}_first_difference"] = [int(diff[0]), int(got[0][diff[0]]), int(got[1][diff[0]]), int(len(diff))]
if "--perf" in sys.argv:
    import _seqgen
    snap = _seqgen.perf_shape(5000, 50_000)
    got = {}
    for knob in (0, 1):
        os.environ["YKPRED_TUNE"] = f"round_batched={knob},round_prof=1"
        pm = pkg.GpuPredicateManager(device=0)
        try:
            pm.load_snapshot(snap)
            pm.evaluate(decisions=True)
            pm.synchronize()
            pm.allocate_round(n=64, apply=False)
            t0 = time.perf_counter()
            got[knob] = pm.allocate_round(apply=False)
            dt = time.perf_counter() - t0
            out[f"perf_shape_{'batched' if knob else 'sequential'}"] = {"asks": int(len(got[knob])), "round_ms": round(dt * 1e3, 2), "allocations_per_sec": round(len(got[knob]) / dt)}
        finally:
            pm.close()
    out["perf_shape_equal"] = bool(np.array_equal(got[0], got[1]))
print(json.dumps(out, indent=1))
