#!/bin/bash
# Round 4, GPU session 15: k_walk_rows (staged ballot rows, index bytes prefetched a group ahead) on the unique-request workload
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/s15
timeout 600 python bench.py --unique-requests --steps 10 --warmup 3 --no-variants --cpu-seconds 0 > gpurun_out/s15/bench_unique.json 2> gpurun_out/s15/bench_unique.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/s15/bench_unique.json'))
print('ms_per_step', d['ms_per_step'], 'verified', d.get('verified'), d.get('verification'))
r=d['roofline']; print({k:r[k] for k in r if k!='kernels'})
print(d.get('kernel_ms') or d.get('kernels'))
PY
tail -3 gpurun_out/s15/bench_unique.err
