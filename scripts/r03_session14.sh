#!/bin/bash
# round 3, GPU session 14: k_combine_wave<ROWS> (class row in registers, member rows written whole) against the column-piece form
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
rm -f gpurun_out/r03_probe2.jsonl
python -c "import importlib; importlib.import_module('yunikorn-k8shim_amd').build_all()" || exit 1
export PROBE_SETS='[
 {"knobs":{"YKPRED_WAVE_ROWS":"1"},"workloads":"own,gang,default","both":true,"check":true},
 {"knobs":{"YKPRED_WAVE_ROWS":"0"},"workloads":"own,gang","both":true},
 {"knobs":{"YKPRED_WAVE_COMBINE_BELOW":"-1"},"workloads":"own","both":true},
 {"knobs":{"YKPRED_WAVE_ROWS":"1","YKPRED_BAND_STEPS":"-1"},"workloads":"own","both":true,"check":true}
]'
timeout 600 python scripts/r03_probe2.py 2>&1 | grep '^{' | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l)
    print(d['knobs'], d['workload'], d['ms_per_step'], d.get('ms_per_step_nodec'), d.get('parity'), 'band_rows', d['band_rows'])
    print('    ', d['kernel_ms'])
    if 'kernel_ms_nodec' in d: print('    nodec', d['kernel_ms_nodec'])"
