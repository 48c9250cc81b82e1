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


# ---- allocation rounds on a node-sharded cluster: the batch protocol of ykpred_allocate_round in reference form -----------------
def round_key(alloc, used):
    """Order key of a node in a bin-pack round of the reference model below: fuller nodes first (smaller key)."""
    return 1.0 - np.asarray(used, dtype=np.float64) / np.asarray(alloc, dtype=np.float64)


def ref_allocate_round_sharded(req, cls, static_ok, alloc, used, node_offset, dist, batch=32, batch_max=256, topk=4):
    """The protocol engine.hip runs for a round on node-sharded engines (allocate_round, batched form), on a model small enough to
    state in a few lines: node n of this shard has capacity alloc[n] of one resource, used[n] of it taken (MUTATED here: the owner
    assumes), ask j requests req[j] and is of class cls[j]; static_ok[c][n] says whether class c may run on node n at all. An ask
    fits a node iff the class may and the request still fits; nodes are tried fullest first, ties by cluster-wide index. Per batch:
    every shard PROPOSES its `topk` best nodes per ask against the state the accepted asks left (k_round_propose) and a bit for every
    (ask of the batch, node the shard proposed in this batch) pair (k_round_cross); one all-gather; every rank REPLAYS the loop over
    the batch: the answer of an ask is the better of the first entry of its merged list that no accepted ask touched (the merged
    list is exact up to the last entry of the first shard list that came back full) and the accepted nodes it passes — its bit, then
    the request against the node's columns after the accepted asks; the batch ends where every known entry is an accepted node that
    is full and the lists were cut. A run of asks with one (class, request) lands on one node while it fits; the owners assume.
    → cluster-wide node per ask, -1 = none; identical on every rank and equal to deciding the asks one after the other over all
    nodes (tests/test_sharding_gloo.py)."""
    world = dist.get_world_size()
    n_asks, out, pos = len(req), np.full(len(req), -1, dtype=np.int64), 0
    while pos < n_asks:
        b = min(batch, n_asks - pos)
        key = round_key(alloc, used)
        lists, number = [], {}  # per ask: [(key, cluster-wide node, alloc, used)]; local node -> its number among the proposed ones
        for j in range(b):
            ok = static_ok[cls[pos + j]] & (alloc - used >= req[pos + j])
            cand = np.flatnonzero(ok)
            cand = cand[np.lexsort((cand, key[cand]))[:topk]]
            lists.append([(float(key[n]), int(node_offset + n), float(alloc[n]), float(used[n])) for n in cand])
            for n in cand:
                number.setdefault(int(n), len(number))
        nodes = np.fromiter(number.keys(), dtype=np.int64, count=len(number))
        bits = [static_ok[cls[pos + j]][nodes] & (alloc[nodes] - used[nodes] >= req[pos + j]) if len(nodes) else np.zeros(0, bool) for j in range(b)]
        gathered = [None] * world
        dist.all_gather_object(gathered, (lists, {int(node_offset + n): d for n, d in number.items()}, bits))
        acc, m = {}, 0  # cluster-wide node -> [key after the accepted asks, alloc, used, owner]
        while m < b:
            ents = sorted((e[0], e[1], r, q) for r, g in enumerate(gathered) for q, e in enumerate(g[0][m]))
            full = [max((e[0], e[1]) for e in g[0][m]) for g in gathered if len(g[0][m]) == topk]
            horizon = min(full) if full else None
            fresh = next((e for e in ents if (horizon is None or (e[0], e[1]) <= horizon) and e[1] not in acc), None)
            passes = [(v[0], u) for u, v in acc.items() if gathered[v[3]][2][m][gathered[v[3]][1][u]] and v[1] - v[2] >= req[pos + m]]
            best = min(passes) if passes else None
            if fresh is None and horizon is not None and not (best is not None and best < horizon):
                break  # (a node behind the lists may fit: only its shard knows)
            if fresh is None and best is None:
                m += 1
                continue
            if fresh is not None and (best is None or (fresh[0], fresh[1]) < best):
                node, owner = fresh[1], fresh[2]
                a_n, u_n = gathered[owner][0][m][fresh[3]][2:4]
            else:
                node = best[1]
                _, a_n, u_n, owner = acc[node]
            holds = (a_n - u_n) // req[pos + m] if req[pos + m] > 0 else 1 << 30
            k = 1
            while k < holds and m + k < b and cls[pos + m + k] == cls[pos + m] and req[pos + m + k] == req[pos + m]:
                k += 1
            u_n = u_n + k * req[pos + m]
            acc[node] = [float(round_key(a_n, u_n)), a_n, u_n, owner]
            out[pos + m:pos + m + k] = int(node)
            local = int(node) - node_offset
            if 0 <= local < len(alloc):
                used[local] += k * req[pos + m]
            m += k
        pos += m
        batch = min(batch_max, max(32, 2 * m + 16))
    return out
