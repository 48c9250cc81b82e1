#!/usr/bin/env python3
"""Latency of the incremental path at configs[2] size (50k nodes x 1M asks): AssumePod → node-row patch → column patch
(ykpred_eval_nodes), with and without a decision refresh, next to a full ykpred_eval. Prints one JSON line."""
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("yunikorn-k8shim_amd")
pm = pkg.GpuPredicateManager()
pm.generate_kwok(seed=0x59554E49 + 2, num_nodes=50_000, num_pods=1_000_000, num_templates=2000, node_affinity=1)
pm.evaluate()
pm.synchronize()


def timed(decisions, n=40, offset=0):
    wall, dev = [], []
    for i in range(n):
        uid = f"pod-{offset + i:07d}"
        node = f"kwok-node-{(7919 * (offset + i)) % 50_000:06d}"
        t0 = time.perf_counter()
        try:
            pm.assume_pod(uid, node)
        except RuntimeError:
            continue  # the ask was generated with a nodeName pin
        k = pm.evaluate_dirty(counts=True, decisions=decisions, profile=True)
        pm.synchronize()
        wall.append((time.perf_counter() - t0) * 1e3)
        assert k == 1
        dev.append(sum(ms for _, ms in pm.timing()["kernels"]))
    return float(np.median(wall)), float(np.median(dev))


def timed_rows(n=40):
    """A new ask of a known template arrives (row appended) / an assumed ask is reported Running (row vacated and refilled)."""
    wall_new, wall_done = [], []
    for i in range(n):
        ask = json.loads(pm.dump_snapshot(pods=[500_000 + 7919 * i], nodes=[]))["pods"][0]
        ask["metadata"].update(uid=f"late-{i}", name=f"late-{i}")
        ask["spec"].pop("nodeName", None)
        t0 = time.perf_counter()
        pm.update_pod(ask)
        pm.evaluate_dirty(counts=True, decisions=True)
        pm.synchronize()
        wall_new.append((time.perf_counter() - t0) * 1e3)
    for i in range(n):
        t0 = time.perf_counter()
        pm.update_pod({"metadata": {"uid": f"late-{i}", "name": f"late-{i}"}, "spec": {}, "status": {"phase": "Succeeded"}})
        pm.evaluate_dirty(counts=True, decisions=True)
        pm.synchronize()
        wall_done.append((time.perf_counter() - t0) * 1e3)
    return float(np.median(wall_new)), float(np.median(wall_done))


def timed_new_label(n=20):
    """A new ask whose nodeSelector uses a (key, value) nobody selected before: dictionary growth in place — the new
    requirement bit is evaluated on all 50k nodes on the host, one label-word column (400 KB) is uploaded, the row is patched."""
    wall = []
    for i in range(n):
        ask = {"metadata": {"uid": f"newlabel-{i}", "name": f"newlabel-{i}", "namespace": "default"},
               "spec": {"nodeSelector": {"kubernetes.io/hostname": f"kwok-node-{1000 + i:06d}"}, "containers": [{"name": "c"}],
                        "tolerations": [{"key": "kwok.x-k8s.io/node", "operator": "Exists", "effect": "NoSchedule"}]}}
        t0 = time.perf_counter()
        pm.update_pod(ask)
        k = pm.evaluate_dirty(counts=True, decisions=True)
        pm.synchronize()
        wall.append((time.perf_counter() - t0) * 1e3)
        assert k >= 0, "dictionary growth must stay incremental"
    return float(np.median(wall)), pm.routing_stats()["dictionary_growths"]


w0, d0 = timed(False, offset=0)
w1, d1 = timed(True, offset=1000)
pm.evaluate()
rn, rd = timed_rows()
nl, growths = timed_new_label()
t0 = time.perf_counter()
for _ in range(5):
    pm.evaluate()
pm.synchronize()
full = (time.perf_counter() - t0) / 5 * 1e3
out = {"workload": "configs[2]: 50k nodes x 1M asks", "assume+column_patch_ms_wall": round(w0, 4),
       "column_patch_kernels_ms": round(d0, 4), "assume+column_patch+decisions_ms_wall": round(w1, 4),
       "decision_refresh_kernels_ms": round(d1, 4), "new_ask_row_patch_ms_wall": round(rn, 4),
       "finished_ask_row_patch_ms_wall": round(rd, 4), "new_label_ask_ms_wall": round(nl, 4), "dictionary_growths": growths,
       "full_eval_ms_wall": round(full, 4)}
pm.close()

# the same AssumePod loop with 10 % of the templates carrying a hard zone-spread constraint: the histograms couple all nodes,
# ykpred_eval_nodes rebuilds them and rewrites the whole rows of the classes whose PreFilter state moved
pm = pkg.GpuPredicateManager()
pm.generate_kwok(seed=0x59554E49 + 2, num_nodes=50_000, num_pods=1_000_000, num_templates=2000, node_affinity=1, spread=1)
pm.evaluate()
pm.synchronize()
# untimed: the generator builds its uid → ask index at the first lookup (clusters fed object by object always have it)
try:
    pm.assume_pod("pod-0999999", "kwok-node-000001")
except RuntimeError:
    pass
pm.evaluate_dirty(counts=True, decisions=False)
pm.synchronize()
wall, incremental = [], 0
for i in range(40):
    uid, node = f"pod-{i:07d}", f"kwok-node-{(7919 * i) % 50_000:06d}"
    t0 = time.perf_counter()
    try:
        pm.assume_pod(uid, node)
    except RuntimeError:
        continue
    k = pm.evaluate_dirty(counts=True, decisions=False)
    pm.synchronize()
    wall.append((time.perf_counter() - t0) * 1e3)
    incremental += k >= 0
t0 = time.perf_counter()
for _ in range(5):
    pm.evaluate()
pm.synchronize()
out["with_hard_spread_constraints"] = {"assume+incremental_ms_wall_median": round(float(np.median(wall)), 4),
                                       "assume+incremental_ms_wall_max": round(float(np.max(wall)), 4),
                                       "incremental_steps": incremental, "steps": len(wall),
                                       "full_eval_ms_wall": round((time.perf_counter() - t0) / 5 * 1e3, 4)}
print(json.dumps(out))
