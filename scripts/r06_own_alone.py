"""Round 6: the own-template population (every ask its own template), the launch stream's kernels timed without and with the decision
branch. Usage on the GPU box: python scripts/r06_own_alone.py [nodes] [pods]"""
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("yunikorn-k8shim_amd")
nodes = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000
pods = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
pm = pkg.GpuPredicateManager(device=0)
pm.generate_kwok(seed=0x59554E49 + 2, num_nodes=nodes, num_pods=pods, num_templates=0, node_affinity=1)
pm.sync()
for dec in (False, True):
    for _ in range(3):
        pm.evaluate(decisions=dec)
    pm.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        pm.evaluate(decisions=dec)
    pm.synchronize()
    ms = (time.perf_counter() - t0) / 10 * 1e3
    kern = {}
    for _ in range(5):
        pm.evaluate(decisions=dec, profile=True)
        for name, v in pm.timing()["kernels"]:
            kern.setdefault(name, []).append(v)
    lay = pm.layout()
    print(f"decisions={dec} ms_per_step={ms:.3f} classes={lay.num_classes} band_rows={lay.band_rows} run_rows={lay.run_rows} sweep_rows={lay.sweep_rows}",
          {k: round(float(np.mean(v)), 4) for k, v in kern.items()}, flush=True)
pm.close()
