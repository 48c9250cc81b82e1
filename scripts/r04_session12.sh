#!/bin/bash
mkdir -p gpurun_out/r04s12
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 900 python -m pytest tests/test_gpu_sequential.py -x -q -m gpu 2>&1 | tail -3
python - <<'PY'
import importlib, sys, time, os
sys.path.insert(0, "tests")
import numpy as np
import _seqgen, _oracle as orc
pkg = importlib.import_module("yunikorn-k8shim_amd")
pm = pkg.GpuPredicateManager()
pm.load_snapshot(_seqgen.perf_shape(5000, 50000))
pm.evaluate(decisions=True); pm.synchronize()
text = pm.dump_snapshot()
pm.allocate_round(n=64, apply=False)
t0 = time.perf_counter(); got = pm.allocate_round(apply=False); t = time.perf_counter() - t0
want = orc.Oracle(text).allocate_sequential()
print("perf shape: %.1f ms, %.0f allocations/s, equal %s" % (t * 1e3, 50000 / t, np.array_equal(got, want)))
pm.generate_kwok(seed=0x59554E49 + 2, num_nodes=50000, num_pods=200000, num_templates=2000, node_affinity=1)
pm.evaluate(decisions=True); pm.synchronize()
asks = np.arange(20000, dtype=np.int32)
pm.allocate_round(asks=asks[:64], apply=False)
t0 = time.perf_counter(); got = pm.allocate_round(asks=asks, apply=False); t = time.perf_counter() - t0
print("50k nodes, 20000 asks of 2000 templates: %.1f ms, %.0f allocations/s, allocated %d" % (t * 1e3, 20000 / t, (got >= 0).sum()))
PY
