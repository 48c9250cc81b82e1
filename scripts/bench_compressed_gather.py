#!/usr/bin/env python3
"""What the class-compressed gather costs per GPU at the configs[3] shard shapes, measured on ONE GPU: evaluate a node shard
of the 50 000-node cluster x 1 M gang asks, collect its class rows, then write `world` slabs — one with the engine's own
writer tables (a peer with the same layout digest), the rest ask by ask through a pod -> class map (a peer that merged
signatures differently). No xGMI in this number: the class rows that would cross the links are `link_MB` per shard."""
import importlib
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("yunikorn-k8shim_amd")
sharding = importlib.import_module("yunikorn-k8shim_amd.sharding")
dev = torch.device("cuda", 0)
out = []
for world in (2, 4, 8):
    ranges = sharding.shard_ranges(50_000, world)
    first, count = ranges[0]
    pm = pkg.GpuPredicateManager()
    pm.generate_kwok(seed=0x59554E49 + 2, num_nodes=count, node_index_offset=first, total_nodes=50_000, num_pods=1_000_000,
                     num_templates=2000, node_affinity=1, gang_size=100)
    pm.set_row_stride(sharding.common_row_stride(ranges))
    pm.set_row_capacity(sharding.common_row_capacity(1_000_000))
    pm.evaluate()
    pm.synchronize()
    lay = pm.layout()
    cls = torch.empty((lay.num_classes, lay.row_stride), dtype=torch.int64, device=dev)
    peer_map = torch.from_numpy(pm.pod_classes()[0].astype(np.int32)).to(dev)
    slabs = torch.empty((world, lay.num_rows, lay.row_stride), dtype=torch.int64, device=dev)
    torch.cuda.synchronize()

    def gather_like():
        pm.collect_class_rows(cls)
        pm.expand_class_rows(cls, slabs[0])
        for g in range(1, world):
            pm.expand_class_rows(cls, slabs[g], pod_class=peer_map)

    for _ in range(2):
        gather_like()
    pm.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        gather_like()
    pm.synchronize()
    ms = (time.perf_counter() - t0) / 5 * 1e3
    t0 = time.perf_counter()
    for _ in range(5):
        pm.evaluate()
    pm.synchronize()
    ev = (time.perf_counter() - t0) / 5 * 1e3
    slab_bytes = lay.num_rows * lay.row_stride * 8
    out.append({"world": world, "shard_nodes": count, "row_stride_words": lay.row_stride, "classes": lay.num_classes,
                "link_MB_per_shard": round(lay.num_classes * lay.row_stride * 8 / 1e6, 2), "slab_GB": round(slab_bytes / 1e9, 3),
                "shard_eval_ms": round(ev, 3), "collect+expand_world_slabs_ms": round(ms, 3),
                "expand_GBps": round(world * slab_bytes / (ms * 1e-3) / 1e9, 1)})
    pm.close()
    del cls, slabs, peer_map
    torch.cuda.empty_cache()
print(json.dumps(out))
