"""Round 6: the decision branch of the unique-request population WITHOUT the bitmap writers beside it (YKPRED_EVAL_SKIP_BITMAP: planes,
rank order, decisions) — what its kernels cost alone. Usage on the GPU box: [YKPRED_TUNE=run_decide=0] python scripts/r06_decisions_alone.py"""
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("yunikorn-k8shim_amd")
pm = pkg.GpuPredicateManager(device=0)
own = len(sys.argv) > 1 and sys.argv[1] == "own"
pm.generate_kwok(seed=0x59554E49 + 2, num_nodes=50_000, num_pods=1_000_000, num_templates=0, node_affinity=1, unique_requests=0 if own else 1)
pm.evaluate()
pm.synchronize()
OUT_COUNTS, OUT_DECISIONS, PROFILE, SKIP = 1 << 1, 1 << 2, 1 << 8, 1 << 12
from importlib import import_module
ffi = import_module("yunikorn-k8shim_amd.predicate_manager")
opts = ffi.OUT_COUNTS | ffi.OUT_DECISIONS | (1 << 12)
for _ in range(3):
    pm._check(pm._L.ykhost_evaluate(pm._h, 1, opts))
pm.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    pm._check(pm._L.ykhost_evaluate(pm._h, 1, opts))
pm.synchronize()
ms = (time.perf_counter() - t0) / 10 * 1e3
kern = {}
for _ in range(5):
    pm._check(pm._L.ykhost_evaluate(pm._h, 1, opts | ffi.EVAL_PROFILE))
    for name, v in pm.timing()["kernels"]:
        kern.setdefault(name, []).append(v)
print(f"decision pass without the bitmap: {ms:.3f} ms", {k: round(float(np.mean(v)), 4) for k, v in kern.items()})
pm.close()
