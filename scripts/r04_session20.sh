#!/bin/bash
# Round 4, GPU session 20: the bench line as the driver produces it (python bench.py, default flags)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/s20
timeout 420 python bench.py > gpurun_out/s20/bench_final.json 2> gpurun_out/s20/bench_final.err
echo "rc=$?"
python - <<'PY'
import json
d=json.load(open('gpurun_out/s20/bench_final.json'))
print('value', d['value'], 'ms', d['ms_per_step'], 'verified', d.get('verified'))
print('roofline', {k:v for k,v in d['roofline'].items()})
for k,v in d.get('variants',{}).items():
    print(k, {a:v.get(a) for a in ('ms_per_step','verified','whole_step_frac')}, (v.get('roofline') or {}).get('kernel'), (v.get('roofline') or {}).get('avg_launch_ms'), (v.get('roofline') or {}).get('frac'), (v.get('roofline') or {}).get('whole_step_frac'), (v.get('roofline') or {}).get('traffic'))
print('alloc', d.get('allocation_round'))
print('e2e', d.get('end_to_end'))
print('cpu', d.get('cpu_baseline'))
PY
tail -2 gpurun_out/s20/bench_final.err
