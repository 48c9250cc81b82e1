"""Ranks of a node-sharded cluster deciding ALLOCATION ROUNDS together (ykhost_allocate_round on a sharded engine: proposals per batch,
one all-gather, the same replay of the sequential loop on every rank, the owners assume) against the oracle's sequential loop over
the WHOLE cluster. Launched by tests/test_gpu_sequential.py through torch.distributed.run.

  SHARD_RCCL_STUB=<tests/c/rccl_stub.cpp built as a shared library>: the ranks share cuda:0 and the engine loads the stub instead of
  librccl (ykpred_comm_use_library) — every line of libykpred's and libykhost's side of the round runs; with >= world GPUs visible and
  no stub: one GPU per rank over RCCL (scripts/scale_check.sh).
Round 1 (apply = 1) takes the first half of the asks, round 2 (apply = 0) the rest on top of what round 1 assumed: both must equal the
oracle's loop over all asks in that order, as GLOBAL node indices, on every rank."""
import importlib
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
pkg = importlib.import_module("yunikorn-k8shim_amd")
sharding = importlib.import_module("yunikorn-k8shim_amd.sharding")
import _oracle as orc  # noqa: E402


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    total_nodes, n_pods, n_templates = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    spread = int(sys.argv[4]) if len(sys.argv) > 4 else 0  # 1: a tenth of the templates carry a hard zone-spread constraint (configs[4]'s mix)
    stub = os.environ.get("SHARD_RCCL_STUB")
    device = 0 if stub else rank
    torch.cuda.set_device(device)
    dist.init_process_group("gloo")
    seed = int(sys.argv[5]) if len(sys.argv) > 5 else 91
    kw = dict(seed=0x59554E49 + seed, num_pods=n_pods, num_templates=n_templates, node_affinity=1, spread=spread)
    ranges = sharding.shard_ranges(total_nodes, world)
    first, count = ranges[rank]
    pm = pkg.GpuPredicateManager(device=device)
    pm.generate_kwok(num_nodes=count, node_index_offset=first, total_nodes=total_nodes, **kw)
    if stub:
        assert pm._P.ykpred_comm_use_library(stub.encode()) == 0
    sharding.attach_communicator(pm, dist, rank, world, first)
    # the whole cluster, CPU only: what the core's loop decides (first fit down the (score, NodeID) order, AssumePod, next ask)
    full = pkg.GpuPredicateManager(device=-1)
    full.generate_kwok(num_nodes=total_nodes, **kw)
    want = orc.Oracle(full.dump_snapshot()).allocate_sequential(prefilter_once=bool(spread))
    full.close()
    half = n_pods // 2
    asks = np.arange(n_pods, dtype=np.int32)
    before = pm.round_stats()
    info0 = pm.round_info()
    t0 = time.perf_counter()
    got1 = pm.allocate_round(asks=asks[:half], apply=True)
    t1 = time.perf_counter()
    got2 = pm.allocate_round(asks=asks[half:], apply=False)
    t2 = time.perf_counter()
    info1 = pm.round_info()
    after = pm.round_stats()
    got = np.concatenate([got1, got2])
    ok = np.array_equal(got, want)
    on_device = after["rounds_on_device"] == before["rounds_on_device"] + 2 and after["asks_one_by_one"] == before["asks_one_by_one"]
    bad = np.flatnonzero(got != want)
    detail = "" if ok else f" first difference at ask {bad[0]}: got {got[bad[0]]} want {want[bad[0]]} ({len(bad)} differ)"
    print(f"rank {rank}/{world} {'rccl-stub' if stub else 'rccl'}{' spread' if spread else ''}: sharded rounds {ok} on_device {on_device} "
          f"({n_pods} asks x {total_nodes} nodes, {int((want >= 0).sum())} allocated on {len(np.unique(want[want >= 0]))} nodes){detail}"
          f" | round 1 incl. the mirror's assumes {half / (t1 - t0):.0f} asks/s, round 2 {(n_pods - half) / (t2 - t1):.0f} asks/s, "
          f"{info1['batches'] - info0['batches']} batches, {info1['exchanges'] - info0['exchanges']} exchanges", flush=True)
    dist.barrier()
    pm.comm_destroy()
    pm.close()
    dist.destroy_process_group()
    sys.exit(0 if (ok and on_device) else 3)


if __name__ == "__main__":
    main()
