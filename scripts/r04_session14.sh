#!/bin/bash
# Round 4, GPU session 14: whole-row k_walk_rows (bit-sliced rank planes) — parity subset, then the unique-request workload
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/s14
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "sorted_walk or slice_writer or unique_request or full_grid" > gpurun_out/s14/pytest.log 2>&1
tail -5 gpurun_out/s14/pytest.log
timeout 600 python bench.py --unique-requests --steps 10 --warmup 3 --no-variants --cpu-seconds 0 > gpurun_out/s14/bench_unique.json 2> gpurun_out/s14/bench_unique.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/s14/bench_unique.json'))
print('ms_per_step', d['ms_per_step'], 'verified', d.get('verified'), d.get('verification'))
r=d['roofline']; print({k:r[k] for k in r if k!='kernels'})
print(r.get('kernels'))
PY
tail -3 gpurun_out/s14/bench_unique.err
