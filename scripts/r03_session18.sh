#!/bin/bash
# round 3, GPU session 18: slice writer on a diet (full batches without guards, NodeName as a template parameter, no per-chunk tests)
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
rm -f gpurun_out/r03_probe2.jsonl
python -c "import importlib; importlib.import_module('yunikorn-k8shim_amd').build_all()" || exit 1
make -C oracle -s || exit 1
timeout 600 python -m pytest tests -m gpu -q --timeout 300 -p no:cacheprovider -x -k "slice_writer or sorted_walk or unique_request or resize" > gpurun_out/pytest_gpu_subset.log 2>&1
echo "pytest rc=$?" >> gpurun_out/pytest_gpu_subset.log
grep -v "^\.\+ *\[" gpurun_out/pytest_gpu_subset.log | tail -8
export PROBE_SETS='[
 {"knobs":{},"workloads":"unique","both":true,"check":true},
 {"knobs":{"YKPRED_SLICE_MODE":"1"},"workloads":"unique","both":true}
]'
timeout 600 python scripts/r03_probe2.py 2>&1 | grep '^{' | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l)
    print(d['knobs'], d['workload'], d['ms_per_step'], d.get('ms_per_step_nodec'), d.get('parity'))
    print('    ', d['kernel_ms'])
    if 'kernel_ms_nodec' in d: print('    nodec', d['kernel_ms_nodec'])"
