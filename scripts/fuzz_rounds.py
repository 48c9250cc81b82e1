#!/usr/bin/env python3
"""Randomised sweep of allocation rounds: the BATCHED form (YKPRED_TUNE round_batched=1: parallel proposals, pair bits, the host's
replay, node-by-node assume) and the sequential kernel (round_batched=0) against the oracle's sequential loop — competing asks of a
few templates on small nodes with taints, selectors, pins, scalar resources, host ports, PodTopologySpread and InterPodAffinity
switched on at random; given orders, template-sorted runs, second rounds on top of the first. Every decision of every round.
Usage: python scripts/fuzz_rounds.py [first] [count]"""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import _oracle as orc  # noqa: E402
import _seqgen  # noqa: E402

pkg = importlib.import_module("yunikorn-k8shim_amd")
first = int(sys.argv[1]) if len(sys.argv) > 1 else 700000
count = int(sys.argv[2]) if len(sys.argv) > 2 else 150
managers = {}
for knob in (1, 0):
    os.environ["YKPRED_TUNE"] = f"round_batched={knob}"
    managers[knob] = pkg.GpuPredicateManager()
bad = 0
for seed in range(first, first + count):
    rng = np.random.default_rng(seed)
    n_nodes, n_pods = int(rng.integers(2, 90)), int(rng.integers(20, 700))
    kw = dict(taints=bool(rng.integers(2)), selectors=bool(rng.integers(2)), pins=bool(rng.integers(2)), scalars=bool(rng.integers(2)),
              spread=rng.integers(4) == 0, ports=rng.integers(4) == 0, ipa=rng.integers(5) == 0)
    snap = _seqgen.competing(seed, n_nodes=n_nodes, n_pods=n_pods, **kw)
    mode = int(rng.integers(3))
    asks = None
    if mode == 1:
        asks = rng.permutation(n_pods)[: max(1, int(n_pods * 0.8))].astype(np.int32)
    elif mode == 2:
        snap["pods"].sort(key=lambda p: (p["metadata"]["labels"]["app"], p["metadata"]["name"]))
    try:
        got = {}
        for knob, pm in managers.items():
            pm.load_snapshot(snap)
            if knob == 1:
                o = orc.Oracle(pm.dump_snapshot())
                want = o.allocate_sequential(pods=asks)
            half = (n_pods if asks is None else len(asks)) // 2
            lst = np.arange(n_pods, dtype=np.int32) if asks is None else asks
            a = pm.allocate_round(asks=lst[:half], apply=True)   # a first round, assumed in the mirror ...
            b = pm.allocate_round(asks=lst[half:], apply=False)  # ... and a second one on top of it
            got[knob] = np.concatenate([a, b])
        ok = np.array_equal(got[1], want) and np.array_equal(got[0], want)
    except Exception as ex:  # noqa: BLE001
        ok = False
        print(f"seed {seed}: {type(ex).__name__}: {ex}", flush=True)
    if not ok:
        bad += 1
        d1 = np.flatnonzero(got.get(1, want) != want)[:3] if 1 in got else []
        d0 = np.flatnonzero(got.get(0, want) != want)[:3] if 0 in got else []
        print(f"seed {seed} MISMATCH nodes={n_nodes} pods={n_pods} {kw} mode={mode} batched differs at {list(d1)} sequential at {list(d0)}", flush=True)
print(f"fuzz_rounds: seeds {first}..{first + count - 1}: {count - bad} ok, {bad} bad", flush=True)
for pm in managers.values():
    pm.close()
sys.exit(1 if bad else 0)
