#!/bin/bash
# round 3, GPU session 13: full GPU suite on the new kernels (k_sig_planes<WPL>, k_slice_desc + k_combine_slices, k_dim_prefix_max,
# k_dim_walk with four words per thread), then the three ask populations
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
rm -f gpurun_out/r03_probe2.jsonl
python -c "import importlib; importlib.import_module('yunikorn-k8shim_amd').build_all()" || exit 1
make -C oracle -s || exit 1
timeout 1200 python -m pytest tests -m gpu -q --timeout 300 -p no:cacheprovider -x > gpurun_out/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
grep -v "^\.\+ *\[" gpurun_out/pytest_gpu.log | tail -30
export PROBE_SETS='[
 {"knobs":{},"workloads":"unique,own,default","both":true,"check":true}
]'
timeout 600 python scripts/r03_probe2.py 2>&1 | grep '^{' | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l)
    print(d['knobs'], d['workload'], d['ms_per_step'], d.get('ms_per_step_nodec'), d.get('parity'))
    print('    ', d['kernel_ms'])
    if 'kernel_ms_nodec' in d: print('    nodec', d['kernel_ms_nodec'])"
