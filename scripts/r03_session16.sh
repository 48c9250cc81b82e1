#!/bin/bash
# round 3, GPU session 16: zone-B chunk list (no workgroups for zone-A chunks) and a SMALL zone B beside the band writer
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
rm -f gpurun_out/r03_probe2.jsonl
python -c "import importlib; importlib.import_module('yunikorn-k8shim_amd').build_all()" || exit 1
make -C oracle -s || exit 1
timeout 600 python -m pytest tests -m gpu -q --timeout 300 -p no:cacheprovider -x -k "incremental or update or expand or gang or empty or dirty or growth or overflow or resident or word_boundary" > gpurun_out/pytest_gpu_subset.log 2>&1
echo "pytest rc=$?" >> gpurun_out/pytest_gpu_subset.log
grep -v "^\.\+ *\[" gpurun_out/pytest_gpu_subset.log | tail -12
export PROBE_SETS='[
 {"knobs":{},"workloads":"default,gang,own","both":true,"check":true},
 {"knobs":{"YKPRED_BESIDE_SMALL":"0"},"workloads":"default,gang","both":true}
]'
timeout 600 python scripts/r03_probe2.py 2>&1 | grep '^{' | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l)
    print(d['knobs'], d['workload'], d['ms_per_step'], d.get('ms_per_step_nodec'), d.get('parity'), 'band_rows', d['band_rows'])
    print('    ', d['kernel_ms'])
    if 'kernel_ms_nodec' in d: print('    nodec', d['kernel_ms_nodec'])"
