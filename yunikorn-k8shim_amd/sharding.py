"""Node-axis sharding across the GPUs of one box: one process per GPU, each holding a contiguous shard of the node table
and the full ask table.

Every (ask, node) pair is independent, so a shard produces its bitmap columns with no data-path collective. The exchanges
of the path live BEHIND THE C ABI (include/ykpred.h: ykpred_comm_init, ykpred_gather_bitmap, ykpred_exchange_decisions, and
the histogram all-reduce inside ykpred_eval) and run over RCCL / xGMI on the engine's streams:

  * bitmaps (BASELINE configs[3]): all-gather of the shard bitmaps into the shard-major layout [G][P][row_stride];
  * decisions: per ask (feasible count, best node) — SUM, MIN order key, MIN global node index;
  * PodTopologySpread / InterPodAffinity: SUM / MAX of the per-shard histograms between the count and min passes.

This module holds the glue a host needs around them (shard geometry, bootstrap of the communicator id, row assembly) and
torch.distributed REFERENCE forms of the same exchanges (`ref_*`): they are what the world-size-2 tests use on machines
where RCCL cannot run (gloo on CPU; two ranks sharing one GPU) and what the product path is checked against.
"""
import numpy as np
import torch

INT32_MAX = 2**31 - 1


# ---- shard geometry ----------------------------------------------------------------------------------------------------
def shard_ranges(total_nodes, world):
    """Contiguous node ranges [(first, count)] — every shard but the last is a multiple of 64 nodes so that bitmap words
    never straddle two shards and the gathered rows concatenate word by word."""
    per = -(-total_nodes // world)
    per = -(-per // 64) * 64
    out, first = [], 0
    for _ in range(world):
        count = max(0, min(per, total_nodes - first))
        out.append((first, count))
        first += count
    return out


def row_stride_words(num_nodes):
    """The engine's automatic row stride for a node count: ceil(N/64) words rounded up to 16 words (128 B), at least 16."""
    return max(16, -(-(-(-num_nodes // 64)) // 16) * 16)


def common_row_stride(ranges):
    return max(row_stride_words(count) for _, count in ranges)


def common_row_capacity(num_pods):
    """Physical bitmap rows every shard allocates: the asks plus room for the writer's layout (band padding, rows appended by
    ask-table patches). Identical on every shard so that the gathered layout [G][rows][row_stride] is regular."""
    return num_pods + num_pods // 64 + 16384


def attach_communicator(pm, dist, rank, world, node_offset):
    """Rank 0 draws the RCCL unique id through the C ABI, the bootstrap group broadcasts it, every rank attaches."""
    box = [pm.comm_unique_id() if rank == 0 else None]
    dist.broadcast_object_list(box, src=0)
    pm.comm_init(box[0], rank, world, node_offset)


def assemble_rows(shard_rows, ranges):
    """Rows in canonical (global) node order from per-shard rows: shard_rows[g] is [n][>= ceil(count_g/64)] uint64."""
    parts = [np.asarray(rows)[:, : -(-count // 64)] for rows, (_, count) in zip(shard_rows, ranges) if count > 0]
    return np.concatenate(parts, axis=1)


# ---- torch.distributed reference forms (tests; fallback when the C-ABI communicator cannot be created) ---------------------
def ref_exchange_decisions(counts, decisions, keys, node_offset, dist):
    """In place: counts → cluster-wide feasible counts; decisions → GLOBAL node index of the best feasible node
    (bin-pack order, ties by global node index) or -1. Same contract as ykpred_exchange_decisions."""
    dist.all_reduce(counts, op=dist.ReduceOp.SUM)
    best_key = keys.clone()
    dist.all_reduce(best_key, op=dist.ReduceOp.MIN)
    mine = (keys == best_key) & (decisions >= 0)
    cand = torch.where(mine, decisions + node_offset, torch.full_like(decisions, INT32_MAX))
    dist.all_reduce(cand, op=dist.ReduceOp.MIN)
    decisions.copy_(torch.where(cand == INT32_MAX, torch.full_like(cand, -1), cand))
    keys.copy_(best_key)
    return counts, decisions


def ref_exchange_spread_histograms(counts, present, dist):
    """SUM of matching pods, MAX of "domain present" over the shards (what ykpred_eval does on an engine with a communicator)."""
    dist.all_reduce(counts, op=dist.ReduceOp.SUM)
    dist.all_reduce(present, op=dist.ReduceOp.MAX)
    return counts, present


def ref_gather_bitmap(local, dist):
    """all_gather of [rows][row_stride] int64 shard bitmaps into [G][rows][row_stride] (same layout as ykpred_gather_bitmap)."""
    world = dist.get_world_size()
    out = torch.empty((world,) + tuple(local.shape), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out.view(-1), local.contiguous().view(-1))
    return out


# names kept for callers of the round-1 module
exchange_decisions = ref_exchange_decisions
exchange_spread_histograms = ref_exchange_spread_histograms


def gathered_row(gathered, row_maps, pod, ranges):
    """Row of ask `pod` in canonical (global) node order from the shard-major gathered layout [G][rows][row_stride].
    `row_maps[g][pod]` is the physical row of the ask in shard g's slab — rows are permuted for the writer (layout.row_of_pod):
    after ykpred_gather_bitmap that is every shard's own map (the gathered maps), after ykpred_gather_bitmap_compressed every
    slab is in the RECEIVING engine's row order, i.e. its pm.row_map() for every g. `ranges` = shard_ranges(...)."""
    parts = [gathered[g, int(row_maps[g][pod]), : -(-count // 64)] for g, (_, count) in enumerate(ranges) if count > 0]
    return torch.cat(parts)
