#!/usr/bin/env python3
"""What the Go manager's mirror costs to fill: every node and pod as the JSON text encoding/json produces, one
ykhost_update_node / ykhost_update_pod per object (SchedulerCache.UpdateNode / UpdatePod hooks; the InitializeState replay of
/root/reference/pkg/cache/context.go:1411-1484) — against the same objects handed over in ONE buffer
(ykhost_update_nodes_batch / ykhost_update_pods_batch). CPU only: a mirror-only handle (device -1) needs no GPU."""
import argparse
import ctypes as C
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("yunikorn-k8shim_amd")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nodes", type=int, default=5000)
    ap.add_argument("--pods", type=int, default=100_000)
    ap.add_argument("--templates", type=int, default=2000)
    ap.add_argument("--device", type=int, default=-1)
    a = ap.parse_args()
    src = pkg.GpuPredicateManager(device=-1)
    src.generate_kwok(seed=0x59554E49 + 2, num_nodes=a.nodes, num_pods=a.pods, num_templates=a.templates, node_affinity=1)
    t0 = time.perf_counter()
    snap = json.loads(src.dump_snapshot())
    src.close()
    node_docs, on_node_docs = [], []
    for n in snap["nodes"]:
        pods = n.pop("pods", [])
        node_docs.append(json.dumps(n).encode())
        for p in pods:
            p.setdefault("spec", {})["nodeName"] = n["metadata"]["name"]
            p.setdefault("status", {})["phase"] = "Running"
            on_node_docs.append(json.dumps(p).encode())
    ask_docs = [json.dumps(p).encode() for p in snap["pods"]]
    t_prep = time.perf_counter() - t0
    out = {"nodes": len(node_docs), "pods_on_nodes": len(on_node_docs), "pending_asks": len(ask_docs),
           "avg_doc_bytes": {"node": sum(map(len, node_docs)) // max(len(node_docs), 1), "pod": sum(map(len, ask_docs)) // max(len(ask_docs), 1)},
           "python_prep_s": round(t_prep, 2)}

    def one_by_one():
        m = pkg.GpuPredicateManager(device=a.device)
        L, h = m._L, m._h
        t = time.perf_counter()
        for d in node_docs:
            L.ykhost_update_node(h, d)
        t_nodes = time.perf_counter() - t
        t = time.perf_counter()
        for d in on_node_docs:
            L.ykhost_update_pod(h, d)
        for d in ask_docs:
            L.ykhost_update_pod(h, d)
        t_pods = time.perf_counter() - t
        return m, t_nodes, t_pods

    def batched():
        m = pkg.GpuPredicateManager(device=a.device)
        L, h = m._L, m._h
        nb, pb = b"\n".join(node_docs), b"\n".join(on_node_docs + ask_docs)
        t = time.perf_counter()
        rn = L.ykhost_update_nodes_batch(h, nb, len(nb))
        t_nodes = time.perf_counter() - t
        t = time.perf_counter()
        rp = L.ykhost_update_pods_batch(h, pb, len(pb))
        t_pods = time.perf_counter() - t
        assert rn == len(node_docs) and rp == len(on_node_docs) + len(ask_docs), (rn, rp, m._L.ykhost_last_error(h))
        return m, t_nodes, t_pods

    for name, fn in (("one_call_per_object", one_by_one), ("batched", batched)):
        m, t_nodes, t_pods = fn()
        t = time.perf_counter()
        if a.device >= 0:
            m.sync()
        else:
            m.encoded_tables()
        t_enc = time.perf_counter() - t
        npods = len(on_node_docs) + len(ask_docs)
        out[name] = {"nodes_ms": round(t_nodes * 1e3, 1), "pods_ms": round(t_pods * 1e3, 1), "us_per_pod": round(t_pods / max(npods, 1) * 1e6, 2),
                     "us_per_node": round(t_nodes / max(len(node_docs), 1) * 1e6, 2), "encode_ms": round(t_enc * 1e3, 1),
                     "asks": m.num_pods, "templates": m.stats()["templates"]}
        m.close()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
