#!/bin/bash
# round 3, GPU session 12+: k_combine_slices v5/v6 — chunk descriptors resolved once per pass (k_slice_desc), cached first plane row, two-pass batches
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
rm -f gpurun_out/r03_probe2.jsonl
python -c "import importlib; importlib.import_module('yunikorn-k8shim_amd').build_all()" || exit 1
export PROBE_SETS='[
 {"knobs":{},"workloads":"unique","both":true,"check":true},
 {"knobs":{"YKPRED_SLICE_MODE":"1"},"workloads":"unique","both":true},
 {"knobs":{"YKPRED_SLICE_CHUNKS":"128"},"workloads":"unique","both":true},
 {"knobs":{"YKPRED_COMBINE_SLICES":"2"},"workloads":"own","check":true}
]'
timeout 600 python scripts/r03_probe2.py 2>&1 | grep '^{' | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l)
    print(d['knobs'], d['workload'], d['ms_per_step'], d.get('ms_per_step_nodec'), d.get('parity'))
    print('    ', d['kernel_ms'])
    if 'kernel_ms_nodec' in d: print('    nodec', d['kernel_ms_nodec'])"
