"""The conflict-resolved allocation rounds of bench.py on their own (no timed evaluation legs): the reference's perf shape, rounds
of 2 000 and 20 000 asks of the main workload (configs[2]), and — with --configs4 — a 20 000-ask round of the configs[4] ask mix
(100 000 nodes, hard spread constraints on a tenth of the templates). Every round is checked against the oracle's sequential loop
(a prefix of the decisions where the oracle would need minutes). Usage on the GPU box: python scripts/bench_rounds.py [--configs4]"""
import importlib
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    pkg = importlib.import_module("yunikorn-k8shim_amd")
    dev = torch.device("cuda", 0)
    out = {}
    big = pkg.GpuPredicateManager(device=0)
    try:
        big.generate_kwok(seed=bench.SEED + 2, num_nodes=50_000, num_pods=1_000_000, num_templates=2000, node_affinity=1)
        out.update(bench.allocation_round_leg(pkg, dev, big_pm=big))
    finally:
        big.close()
    if "--configs4" in sys.argv:
        pm = pkg.GpuPredicateManager(device=0)
        try:
            pm.generate_kwok(seed=bench.SEED + 4, num_nodes=100_000, num_pods=1_000_000, num_templates=2000, node_affinity=1, spread=1)
            out["configs4_shape_round"] = bench.device_rounds(pm, [20_000], 300)[0]
        finally:
            pm.close()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
