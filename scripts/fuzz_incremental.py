#!/usr/bin/env python3
"""Randomised sweep of the INCREMENTAL paths (GPU vs oracle): random sequences of SchedulerCache operations — new asks
(known and new templates), deletions, AssumePod / ForgetPod, binds completing, node object edits, node removal and
re-addition — each followed by evaluate_dirty(); after every step every live ask row, count and decision must equal the
oracle's on the mirror's own snapshot dump, and at the end a full evaluation must reproduce the patched bitmap.
Usage: python scripts/fuzz_incremental.py [first_seed] [count] [steps]"""
import copy
import importlib
import json
import os
import random
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import _gen  # noqa: E402
import _oracle as orc  # noqa: E402

pkg = importlib.import_module("yunikorn-k8shim_amd")
first = int(sys.argv[1]) if len(sys.argv) > 1 else 700000
count = int(sys.argv[2]) if len(sys.argv) > 2 else 40
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 30
pm = pkg.GpuPredicateManager()


def unpack(bitmap, n):
    return np.unpackbits(bitmap.view(np.uint8), axis=1, bitorder="little")[:, :n]


def check(tag, decisions):
    snap = json.loads(pm.dump_snapshot())
    if not snap["pods"] or not snap["nodes"]:
        return True
    o = orc.Oracle(snap)
    want = o.eval_grid(threads=16)
    idx = [pm.pod_index(p["metadata"]["uid"]) for p in snap["pods"]]
    lay = pm.layout()
    if min(idx) < 0 or lay.num_pods != pm.num_pods:
        print(f"  {tag}: row bookkeeping broken ({idx[:5]}, {lay.num_pods} vs {pm.num_pods})")
        return False
    got = unpack(pm.read_bitmap(), lay.num_nodes)[idx]
    if not np.array_equal(got, want):
        bad = np.argwhere(got != want)
        print(f"  {tag}: {len(bad)} differing bits, first {bad[0].tolist()} uid={snap['pods'][bad[0][0]]['metadata']['uid']}")
        return False
    if not np.array_equal(pm.read_counts()[idx], want.sum(axis=1)):
        print(f"  {tag}: counts differ")
        return False
    if decisions:
        dec = pm.read_decisions()[idx]
        for k in range(len(idx)):
            if o.decide(k) != (int(want[k].sum()), int(dec[k])):
                print(f"  {tag}: decision of {snap['pods'][k]['metadata']['uid']} differs: {o.decide(k)} vs {int(dec[k])}")
                return False
    return True


failures = 0
for seed in range(first, first + count):
    rng = random.Random(seed)
    topo = seed % 3 == 0
    snap = _gen.random_snapshot(seed, n_nodes=rng.randint(3, 200), n_pods=rng.randint(2, 60), scalars=bool(seed % 2), spread=topo, interpod=topo)
    extra = _gen.random_snapshot(seed + 10**6, n_nodes=4, n_pods=12, scalars=bool(seed % 2), spread=topo, interpod=topo)["pods"]
    try:
        if os.environ.get("FUZZ_TRACE"):
            print(f"seed {seed} load + full evaluation ({len(snap['nodes'])} nodes, {len(snap['pods'])} pods)", flush=True)
        pm.load_snapshot(snap)
        pm.evaluate()
        node_objs = {n["metadata"]["name"]: {k: v for k, v in n.items() if k != "pods"} for n in snap["nodes"] if n["metadata"]["name"]}
        removed_nodes = {}
        live = {p["metadata"]["uid"]: p for p in snap["pods"]}
        serial = 0
        ok = True
        for step in range(steps):
            op = rng.choice(["new_known", "new_known", "new_template", "delete", "assume", "assume", "forget", "running", "edit_node",
                             "remove_node", "readd_node"])
            names = sorted(node_objs)
            if op == "new_known" and live:
                src = copy.deepcopy(rng.choice(list(live.values())))
                serial += 1
                src["metadata"].update(uid=f"n{serial}", name=f"n{serial}")
                src["spec"].pop("nodeName", None)
                src.pop("status", None)
                pm.update_pod(src)
                live[src["metadata"]["uid"]] = src
            elif op == "new_template" and extra:
                src = copy.deepcopy(extra.pop())
                serial += 1
                src["metadata"].update(uid=f"t{serial}", name=f"t{serial}")
                src["spec"].pop("nodeName", None)
                pm.update_pod(src)
                live[src["metadata"]["uid"]] = src
            elif op == "delete" and live:
                uid = rng.choice(sorted(live))
                pm.remove_pod(uid)
                live.pop(uid)
            elif op == "assume" and live and names:
                uid = rng.choice(sorted(live))
                st = pm.pod_state(uid)
                if st and st["ask"]:
                    pm.assume_pod(uid, rng.choice(names))
            elif op == "forget" and live:
                cands = [u for u in sorted(live) if (pm.pod_state(u) or {}).get("assumed")]
                if cands:
                    pm.forget_pod(rng.choice(cands))
            elif op == "running" and live:
                cands = [u for u in sorted(live) if (pm.pod_state(u) or {}).get("assumed")]
                if cands:
                    uid = rng.choice(cands)
                    pm.update_pod(dict(live[uid], status={"phase": "Running"}))
                    live.pop(uid)
            elif op == "edit_node" and names:
                n = copy.deepcopy(node_objs[rng.choice(names)])
                n.setdefault("spec", {})
                n.setdefault("status", {}).setdefault("allocatable", {})
                kind = rng.randrange(4)
                if kind == 0:
                    n["spec"]["unschedulable"] = not n["spec"].get("unschedulable", False)
                elif kind == 1:
                    n["status"]["allocatable"]["cpu"] = str(rng.choice([0, 1, 4, 64]))
                elif kind == 2:
                    n["spec"]["taints"] = copy.deepcopy(node_objs[rng.choice(names)].get("spec", {}).get("taints", []))
                else:
                    n["metadata"]["labels"] = copy.deepcopy(node_objs[rng.choice(names)]["metadata"].get("labels", {}))
                pm.update_node(n)
                node_objs[n["metadata"]["name"]] = n
            elif op == "remove_node" and len(names) > 1:
                name = rng.choice(names)
                pm.remove_node(name)
                removed_nodes[name] = node_objs.pop(name)
            elif op == "readd_node" and removed_nodes:
                name = rng.choice(sorted(removed_nodes))
                n = removed_nodes.pop(name)
                pm.update_node(n)
                node_objs[name] = n
            else:
                continue
            dec = bool(rng.getrandbits(1))
            if os.environ.get("FUZZ_TRACE"):
                print(f"seed {seed} step {step} {op} decisions={dec}", flush=True)
            pm.evaluate_dirty(decisions=dec)
            if not check(f"seed={seed} step={step} op={op}", dec):
                ok = False
                break
        if ok:
            before = pm.read_bitmap().copy()
            pm.evaluate()
            if not np.array_equal(before, pm.read_bitmap()) or not check(f"seed={seed} final", True):
                print(f"  seed={seed}: full evaluation differs from the patched state")
                ok = False
        failures += 0 if ok else 1
    except RuntimeError as e:
        if "engine limit" in str(e) or "not supported" in str(e) or "unsupported" in str(e):
            continue
        failures += 1
        print(f"ERROR seed={seed}: {e}", flush=True)
print(f"fuzz_incremental: {count} clusters x {steps} steps from seed {first}, {failures} failures")
sys.exit(1 if failures else 0)
