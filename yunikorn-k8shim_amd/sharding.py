"""Node-axis sharding across the GPUs of one box (one process per GPU, torch.distributed over RCCL/xGMI).

Every (pod,node) pair is independent for NodeResourcesFit / TaintToleration / NodeAffinity / NodeUnschedulable /
NodeName, so each rank evaluates all asks against its own contiguous shard of nodes with no data-path collective.
What the path does exchange:
  * decisions: per pod (feasible count, best node) — a SUM and two MIN all-reduces of P-element vectors
    (exchange_decisions), ≈16 MB per rank for 1 M asks;
  * optionally the shard bitmaps themselves (config 4 of BASELINE.json): all_gather_into_tensor of
    [P][row_stride] u64 per rank into the shard-major layout [G][P][row_stride].
torch is plumbing here (device tensors + the collective); the verdicts come from the engine.
"""
import time

import torch

INT32_MAX = 2**31 - 1


def exchange_decisions(counts, decisions, keys, node_offset, dist):
    """In place: counts → cluster-wide feasible counts; decisions → GLOBAL node index of the best feasible node
    (bin-pack order, ties by global node index) or -1.

    counts int32[P], decisions int32[P] (local node index or -1), keys int64[P] (order key of the local best,
    INT64_MAX if none) are this rank's outputs of ykpred_eval; node_offset = global index of local node 0."""
    dist.all_reduce(counts, op=dist.ReduceOp.SUM)
    best_key = keys.clone()
    dist.all_reduce(best_key, op=dist.ReduceOp.MIN)
    mine = (keys == best_key) & (decisions >= 0)
    cand = torch.where(mine, decisions + node_offset, torch.full_like(decisions, INT32_MAX))
    dist.all_reduce(cand, op=dist.ReduceOp.MIN)
    decisions.copy_(torch.where(cand == INT32_MAX, torch.full_like(cand, -1), cand))
    return counts, decisions


def exchange_spread_histograms(counts, present, dist):
    """PodTopologySpread on a node-sharded cluster: every shard builds partial per-(constraint, domain) histograms
    over ITS nodes (ykpred_eval with YKPRED_EVAL_SPREAD_COUNT_ONLY); the cluster-wide PreFilter state is their SUM
    (matching pods) and MAX (domain present on some eligible node). Afterwards each shard evaluates with
    YKPRED_EVAL_SPREAD_COUNTS_READY. Requires identical topology-domain dictionaries on all shards (the host side
    builds them from the whole cluster). In place; KBs of traffic."""
    dist.all_reduce(counts, op=dist.ReduceOp.SUM)
    dist.all_reduce(present, op=dist.ReduceOp.MAX)
    return counts, present


def gathered_row(gathered, pod, num_shards, row_words):
    """Row of `pod` in the canonical node order from the shard-major gathered layout [G][P][row_stride]."""
    return torch.cat([gathered[g, pod, :row_words] for g in range(num_shards)])


def time_bitmap_allgather(pm, dist, dev, repeats=3):
    """Evaluates into a torch-owned bitmap and times the RCCL all-gather of the shard bitmaps (BASELINE config 4)."""
    lay = pm.layout()
    world = dist.get_world_size()
    local = torch.empty((lay.num_pods, lay.row_stride), dtype=torch.int64, device=dev)
    pm.evaluate_into(bitmap=local, stream=torch.cuda.current_stream(dev).cuda_stream)
    out = torch.empty((world, lay.num_pods, lay.row_stride), dtype=torch.int64, device=dev)
    torch.cuda.synchronize(dev)
    dist.barrier()
    times = []
    for _ in range(repeats):
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        dist.all_gather_into_tensor(out.view(-1), local.view(-1))
        torch.cuda.synchronize(dev)
        times.append(time.perf_counter() - t0)
    t = torch.tensor([min(times)], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    nbytes = local.numel() * 8
    return {"shard_bytes": nbytes, "ms": float(t.item()) * 1e3, "recv_GBps_per_gpu": nbytes * (world - 1) / float(t.item()) / 1e9,
            "layout": "[G][P][row_stride] u64 (shard-major)"}
