import importlib, sys, time, os
sys.path.insert(0, ".")
import numpy as np, torch
pkg = importlib.import_module("yunikorn-k8shim_amd")
tpl = int(os.environ.get("TPL", "2000"))
pm = pkg.GpuPredicateManager()
pm.generate_kwok(seed=0x59554E49 + 2, num_nodes=50_000, num_pods=1_000_000, num_templates=tpl, node_affinity=1)
for dec in (False, True):
    for _ in range(3):
        pm.evaluate(decisions=dec)
    pm.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        pm.evaluate(decisions=dec)
    pm.synchronize()
    ms = (time.perf_counter() - t0) / 20 * 1e3
    ks = {}
    for _ in range(5):
        pm.evaluate(decisions=dec, profile=True)
        for k, v in pm.timing()["kernels"]:
            ks.setdefault(k, []).append(v)
    print("tpl", tpl, "decisions", dec, "ms/step %.3f" % ms, {k: round(float(np.mean(v)), 4) for k, v in ks.items() if np.mean(v) > 0.03}, flush=True)
