#!/usr/bin/env python3
"""Latency of the per-pair drop-in path (PredicateManager.Predicates → ykpred_query) and throughput of batched queries
at configs[2] size. Prints one JSON line."""
import importlib, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("yunikorn-k8shim_amd")
pm = pkg.GpuPredicateManager()
pm.generate_kwok(seed=0x59554E49 + 2, num_nodes=50_000, num_pods=1_000_000, num_templates=2000, node_affinity=1)
pm.sync()
rng = np.random.default_rng(1)
for _ in range(20):
    pm.predicates(int(rng.integers(0, 1_000_000)), int(rng.integers(0, 50_000)), True)
t0 = time.perf_counter()
n = 2000
for _ in range(n):
    pm.predicates(int(rng.integers(0, 1_000_000)), int(rng.integers(0, 50_000)), True)
single_us = (time.perf_counter() - t0) / n * 1e6
out = {"Predicates_call_us_random_pod_each_call": round(single_us, 1)}
# the core's pattern: one ask tried on many nodes in a row
t0 = time.perf_counter()
calls = 0
for pod in range(100, 120):
    for node in range(0, 50_000, 25):
        pm.predicates(pod, node, True)
        calls += 1
out["Predicates_call_us_one_ask_over_2000_nodes"] = round((time.perf_counter() - t0) / calls * 1e6, 2)
for batch in (1_000, 100_000, 4_000_000):
    p = rng.integers(0, 1_000_000, batch).astype(np.int32)
    q = rng.integers(0, 50_000, batch).astype(np.int32)
    pm.query(p, q)
    t0 = time.perf_counter()
    pm.query(p, q)
    dt = time.perf_counter() - t0
    out[f"batch_{batch}_pairs_per_s"] = round(batch / dt)
print(json.dumps(out))
