#!/usr/bin/env python3
"""bench.py — pod×node predicate evaluations/s of the MI355X engine on BASELINE.json's metric config.

A "step" is ONE pass of the hot path over one snapshot: every pending ask against every node through the whole
default allocation-phase Filter set (NodeUnschedulable, NodeName, TaintToleration, NodeAffinity, NodeResourcesFit;
PodTopologySpread skips — no pod carries hard constraints), producing the P×N feasibility bitmap, the per-pod
feasible-node count and the per-pod bin-pack decision. Tables are resident in HBM when the timed region starts.

  N = 1   workload = configs[2]: 50 000 nodes × 1 000 000 pods (KWOK-style synthetic, 2 000 pod templates).
  N > 1   node-axis sharding, WEAK scaling: every rank holds its own 50 000-node shard of a N·50 000-node cluster
          and the same 1 M asks. The only exchange the path needs for decisions is per-pod (count, best node):
          one SUM and two MIN all-reduces of P-element vectors over RCCL — not the bitmap (`--gather-bitmap`
          adds the config-4 style all-gather of shard bitmaps and is reported separately, never in `value`).

Prints ONE JSON line (rank 0). `roofline` describes k_combine, the kernel that writes the bitmap.
"""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
SEED = 0x59554E49  # "YUNI"


def baseline_metric():
    """BASELINE.json's metric string (evals/sec is `value`, decisions/sec rides along as `decisions_per_sec`)."""
    try:
        with open(os.path.join(ROOT, "BASELINE.json")) as f:
            return json.load(f)["metric"]
    except (OSError, KeyError, ValueError):
        return "pod×node predicate evals/sec + decisions/sec, 50k nodes × 1M pods"


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--nodes", type=int, default=50_000, help="nodes per GPU")
    ap.add_argument("--pods", type=int, default=1_000_000)
    ap.add_argument("--templates", type=int, default=2000, help="distinct pod templates (0 = every pod draws its own)")
    ap.add_argument("--unique-requests", action="store_true", help="adversarial: a distinct cpu request per pod")
    ap.add_argument("--no-affinity", action="store_true", help="configs[1] plugin mix (no nodeSelector/affinity on pods)")
    ap.add_argument("--spread", action="store_true", help="configs[4] plugin mix: 10 %% of the templates carry a hard zone-spread constraint")
    ap.add_argument("--gang", type=int, default=0, help="configs[3] shape: asks are gang placeholders, this many identical members per group")
    ap.add_argument("--direct", action="store_true", help="time the per-pair kernel instead of the plane/class path")
    ap.add_argument("--gather-bitmap", action="store_true", help="N>1: also all-gather the shard bitmaps (reported separately)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the cpu_baseline leg (0 = skip)")
    ap.add_argument("--variant", type=int, default=0, help="k_combine store flavour (0 dwordx4, 1 dwordx2, 2/3 = non-temporal)")
    ap.add_argument("--profile-steps", type=int, default=5, help="steps run with per-kernel HIP events for `roofline`")
    return ap.parse_args()


def cpu_baseline(pm, budget_s, seed):
    """The reference's path cannot run here (Go); this times the oracle — a per-pair, object-model restatement of
    Predicates() — single-threaded (the core drives Predicates serially) on a bounded sample of the same workload."""
    if budget_s <= 0:
        return None
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import _oracle as orc
    rng = np.random.default_rng(seed)
    n_nodes = min(pm.num_nodes, 4096)
    nodes = np.sort(rng.choice(pm.num_nodes, n_nodes, replace=False)).astype(np.int32)
    pods = np.sort(rng.choice(pm.num_pods, min(pm.num_pods, 16384), replace=False)).astype(np.int32)
    o = orc.Oracle(pm.dump_snapshot(pods=pods, nodes=nodes))
    t0 = time.perf_counter()
    o.eval_grid(pods=np.arange(8, dtype=np.int32), threads=1)
    per_pair = (time.perf_counter() - t0) / (8 * n_nodes)
    use = int(max(8, min(len(pods), budget_s / max(per_pair * n_nodes, 1e-9))))
    t0 = time.perf_counter()
    o.eval_grid(pods=np.arange(use, dtype=np.int32), threads=1)
    dt = time.perf_counter() - t0
    # second, stronger figure: the same per-pair port spread over every host core, on a sample sized for
    # about a quarter of the budget
    cores = os.cpu_count() or 1
    use_mt = int(min(len(pods), max(use, use * cores // 4)))
    t0 = time.perf_counter()
    o.eval_grid(pods=np.arange(use_mt, dtype=np.int32), threads=cores)
    dt_mt = time.perf_counter() - t0
    out = {"value": use * n_nodes / dt, "unit": "evals/s", "cores": 1, "kind": "port",
           "sample": f"{use} sampled pods x {n_nodes} sampled nodes of the same workload ({use * n_nodes} Predicates() calls, {dt:.1f} s)",
           "all_cores": {"value": use_mt * n_nodes / dt_mt, "cores": cores,
                         "sample": f"{use_mt} pods x {n_nodes} nodes, {dt_mt:.1f} s"}}
    # third figure: the strongest CPU formulation we know without the GPU path's planes / classes — the ENCODED tables
    # evaluated per pair with bitmask compares on every core (oracle/soa_cpu.c, checked against the oracle in tests)
    try:
        import _soa_cpu
        mirror = importlib.import_module("yunikorn-k8shim_amd").GpuPredicateManager(device=-1)
        mirror.load_snapshot(pm.dump_snapshot(pods=pods, nodes=nodes))
        tables = mirror.encoded_tables()
        mirror.close()
        if not tables["KD"] and not tables["spread_constraints"]:
            prepared = _soa_cpu.prepare(tables)
            _soa_cpu.run(prepared, orc.ALL, orc.ALL, threads=cores)
            reps, t0 = 0, time.perf_counter()
            while time.perf_counter() - t0 < max(1.0, budget_s / 6):
                _soa_cpu.run(prepared, orc.ALL, orc.ALL, threads=cores)
                reps += 1
            dt_soa = time.perf_counter() - t0
            out["soa_all_cores"] = {"value": reps * tables["P"] * tables["N"] / dt_soa, "cores": cores, "kind": "encoded tables, per pair, OpenMP",
                                    "sample": f"{reps} passes over {tables['P']} pods x {tables['N']} nodes, {dt_soa:.1f} s"}
    except Exception as exc:  # noqa: BLE001 - the extra figure must never break the bench line
        out["soa_all_cores"] = {"error": str(exc)}
    return out


def main():
    a = parse_args()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    # BENCH_BACKEND=gloo lets the N>1 path be exercised on a single-GPU box (all ranks share cuda:0); the driver's
    # multi-GPU runs use the default: nccl = RCCL over xGMI, one rank per GPU.
    backend = os.environ.get("BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank = local_rank % max(torch.cuda.device_count(), 1)
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    pkg = importlib.import_module("yunikorn-k8shim_amd")
    from importlib import import_module
    shard = import_module("yunikorn-k8shim_amd.sharding")
    pm = pkg.GpuPredicateManager(device=local_rank)
    t_gen = time.perf_counter()
    pm.generate_kwok(seed=SEED + 2, num_nodes=a.nodes, num_pods=a.pods, num_templates=a.templates,
                     node_affinity=0 if a.no_affinity else 1, unique_requests=1 if a.unique_requests else 0,
                     node_index_offset=rank * a.nodes, spread=1 if a.spread else 0, gang_size=a.gang)
    pm.sync()
    t_gen = time.perf_counter() - t_gen
    P, N = pm.num_pods, pm.num_nodes

    # caller-owned outputs (torch tensors) so that the exchange step can run on them. Two sets: for N>1 the decision
    # exchange of step k (RCCL all-reduces on a side stream) overlaps the evaluation of step k+1 on the main stream.
    nbuf = 2 if world > 1 else 1
    outs = [(torch.empty(P, dtype=torch.int32, device=dev), torch.empty(P, dtype=torch.int32, device=dev),
             torch.empty(P, dtype=torch.int64, device=dev)) for _ in range(nbuf)]
    # A dedicated (non-default) stream: the engine launches on the stream it is handed, and handle 0 — torch's default
    # stream — would mean "use the engine's own stream", which the events below could not order against.
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    comm = torch.cuda.Stream(device=dev) if world > 1 else None
    exchanged = [torch.cuda.Event() for _ in range(nbuf)]  # exchange that last used buffer set b has finished
    step_no = [0]

    def step(profile=False):
        b = step_no[0] % nbuf
        step_no[0] += 1
        counts, decisions, keys = outs[b]
        if world > 1:
            stream.wait_event(exchanged[b])  # the eval below overwrites set b: its previous exchange must be done
        pm.evaluate_into(counts=counts, decisions=decisions, keys=keys if world > 1 else None, stream=stream.cuda_stream,
                         profile=profile, direct=a.direct, variant=a.variant)
        if world > 1:
            evaluated = torch.cuda.Event()
            evaluated.record(stream)
            comm.wait_event(evaluated)
            with torch.cuda.stream(comm):
                shard.exchange_decisions(counts, decisions, keys, rank * a.nodes, dist)
                exchanged[b].record(comm)

    def drain():
        if comm is not None:
            stream.wait_stream(comm)
        torch.cuda.synchronize(dev)

    for _ in range(a.warmup):
        step()
    drain()
    if dist:
        dist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    drain()
    if dist:
        dist.barrier()
    torch.cuda.synchronize(dev)
    elapsed = time.perf_counter() - t0
    if dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    lay = pm.layout()
    # ---- roofline of the dominant kernel (HIP events on the launch stream, bracketing each kernel)
    kern = {}
    for _ in range(max(a.profile_steps, 0)):
        step(profile=True)
        drain()
        for name, ms in pm.timing()["kernels"]:
            kern.setdefault(name, []).append(ms)
    dom = "k_direct" if a.direct else "k_combine"
    st = pm.stats()
    b_node = 8 * 2 * st["R"] + 4 + 4 + 4 + 8 * st["KT"] + 8 * st["W"]
    bitmap_bytes = P * lay.row_words * 8
    # ALGORITHMIC bytes of one launch: bitmap written once + node table + pod table read once
    algo_bytes = bitmap_bytes + N * b_node + P * (4 + 4 + 4) + lay.num_classes * 4 * 4 + lay.plane_rows * lay.row_words * 8
    roof = None
    if dom in kern:
        avg_ms = float(np.mean(kern[dom]))
        achieved = algo_bytes / (avg_ms * 1e-3) / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic_r01.json")
        if os.path.exists(tpath):
            tj = json.load(open(tpath))
            if tj.get("pods") == P and tj.get("nodes") == N and tj.get("kernel") == dom:
                traffic = tj.get("hbm_bytes_per_launch")
        roof = {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "avg_launch_ms": round(avg_ms, 4),
                "algorithmic_bytes": int(algo_bytes)}

    gather = None
    if a.gather_bitmap and dist:
        gather = shard.time_bitmap_allgather(pm, dist, dev)

    cpu = cpu_baseline(pm, a.cpu_seconds, SEED) if rank == 0 else None
    if rank == 0:
        evals = float(P) * float(N) * world * a.steps
        out = {
            "metric": baseline_metric(), "value": evals / elapsed, "unit": "evals/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": elapsed / a.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "int64", "data": "synthetic",
            "config": {"workload": ("configs[2]: 50k nodes x 1M pods, NodeResourcesFit + TaintToleration + NodeAffinity"
                                    if (N, P, a.no_affinity, a.spread, a.gang) == (50_000, 1_000_000, False, False, 0) else
                                    f"{N} nodes/GPU x {P} pods, affinity={'off' if a.no_affinity else 'on'}"),
                       "nodes_per_gpu": N, "pods": P, "templates": a.templates, "pod_classes": lay.num_classes,
                       "signature_planes": lay.plane_rows, "unique_requests": bool(a.unique_requests), "spread": bool(a.spread), "gang_size": a.gang,
                       "path": "direct" if a.direct else "planes+combine",
                       "parallelism": "single GPU" if world == 1 else
                       f"node-axis shards x{world}; per-pod decision all-reduces of step k overlap the evaluation of step k+1"},
            "decisions_per_sec": float(P) * a.steps / elapsed,
            "roofline": roof, "cpu_baseline": cpu,
            "kernel_ms": {k: round(float(np.mean(v)), 4) for k, v in kern.items()},
            "host_setup_s": round(t_gen, 2), "encode_ms": round(st["encode_us"] / 1e3, 1),
        }
        if gather:
            out["bitmap_allgather"] = gather
        print(json.dumps(out))
    pm.close()
    if dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
