#!/usr/bin/env python3
"""Where does the time of the small-class populations go? Own-template-per-ask and unique-request-vector workloads at 50 k x 1 M,
evaluated with and without the decision branch (the aux-stream kernels run beside the writers and compete with them), per-kernel
HIP-event times and wall time per step."""
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("yunikorn-k8shim_amd")
SEED = 0x59554E49
out = []
for name, kw in (("own_template_per_ask", dict(num_templates=0)), ("unique_request_vectors", dict(num_templates=0, unique_requests=1)),
                 ("default", dict(num_templates=2000))):
    pm = pkg.GpuPredicateManager()
    pm.generate_kwok(seed=SEED + 2, num_nodes=50_000, num_pods=1_000_000, node_affinity=1, **kw)
    pm.sync()
    for decisions in ((True, False) if os.environ.get("PROBE_BOTH") else (True,)):
        for _ in range(2):
            pm.evaluate(decisions=decisions)
        pm.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            pm.evaluate(decisions=decisions)
        pm.synchronize()
        ms = (time.perf_counter() - t0) / 5 * 1e3
        kern = {}
        for _ in range(3):
            pm.evaluate(decisions=decisions, profile=True)
            for k, v in pm.timing()["kernels"]:
                kern.setdefault(k, []).append(v)
        lay = pm.layout()
        rec = {"knobs": {k: v for k, v in os.environ.items() if k.startswith("YKPRED_")}, "workload": name, "decisions": decisions, "ms_per_step": round(ms, 4), "classes": lay.num_classes, "band_rows": lay.band_rows,
               "rows": lay.num_rows, "planes": lay.plane_rows, "index_rows": lay.index_rows, "band_steps": lay.band_steps, "kernel_ms": {k: round(float(np.mean(v)), 4) for k, v in kern.items()}}
        out.append(rec)
        print(json.dumps(rec), flush=True)
    pm.close()
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
with open(os.path.join(ROOT, "gpurun_out", "r03_variant_probe.jsonl"), "a") as f:
    for rec in out:
        f.write(json.dumps(rec) + "\n")
