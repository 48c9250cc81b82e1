#!/usr/bin/env python3
"""configs[3] shard shapes on ONE GPU: how fast is a slab of the gathered bitmap written (a) by this engine's own writer tables
(band writer + chunk writer: what a peer with MY layout digest costs) and (b) row by row through a peer's ask -> class map
(k_expand_by_row: a peer whose class partition differs), as a function of the band height? Shard evaluation time alongside."""
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
for world in (8, 4, 2):
    ranges = sharding.shard_ranges(50_000, world)
    first, count = ranges[0]
    for steps in ("0", "128", "32", "16", "8"):
        os.environ["YKPRED_BAND_STEPS"] = steps
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
        slab = torch.empty((lay.num_rows, lay.row_stride), dtype=torch.int64, device=dev)
        pm.collect_class_rows(cls)
        torch.cuda.synchronize()

        def timed(fn, n=8):
            fn()
            pm.synchronize()
            t0 = time.perf_counter()
            for _ in range(n):
                fn()
            pm.synchronize()
            return (time.perf_counter() - t0) / n * 1e3

        own = timed(lambda: pm.expand_class_rows(cls, slab))
        by_row = timed(lambda: pm.expand_class_rows(cls, slab, pod_class=peer_map))
        ev = timed(lambda: pm.evaluate(), 5)
        slab_bytes = lay.num_rows * lay.row_stride * 8
        out.append({"world": world, "band_steps": steps, "shard_nodes": count, "row_bytes": lay.row_stride * 8, "classes": lay.num_classes,
                    "band_rows": lay.band_rows, "rows": lay.num_rows, "slab_GB": round(slab_bytes / 1e9, 3), "shard_eval_ms": round(ev, 3),
                    "own_tables_ms": round(own, 3), "own_tables_GBps": round(slab_bytes / own / 1e6, 1),
                    "by_row_ms": round(by_row, 3), "by_row_GBps": round(slab_bytes / by_row / 1e6, 1)})
        print(json.dumps(out[-1]), flush=True)
        pm.close()
        del cls, slab, peer_map
        torch.cuda.empty_cache()
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "r03_shard_shapes.json"), "w"), indent=1)
