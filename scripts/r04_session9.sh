#!/bin/bash
# Round 4, GPU session 9: k_walk_rows knobs on the unique-request population (store throttle, chunks per workgroup)
mkdir -p gpurun_out/r04s9
cd "$GRAFT_REPO_ROOT" || exit 1
run() {
  name=$1; shift
  env YKPRED_TUNE="$1" timeout 600 python bench.py --templates 0 --unique-requests --no-variants --no-verify --cpu-seconds 0 --steps 10 --profile-steps 3 > gpurun_out/r04s9/$name.json 2> gpurun_out/r04s9/$name.err
  python - gpurun_out/r04s9/$name.json "$name" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    k=d['kernel_ms']; print(f"{sys.argv[2]:32s} step {d['ms_per_step']:.3f} ms  desc {k.get('k_slice_desc')} walk {k.get('k_walk_rows')} general {k.get('k_combine')}")
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "slice_writer or sorted_walk or unique_request" 2>&1 | tail -2
run base "walk_chunks=2048"
run dbg "walk_debug=1"
grep "k_walk_rows cycles" gpurun_out/r04s9/dbg.err | tail -1
run slots6 "walk_run_slots=6,walk_buffers=10"
run slots16 "walk_run_slots=16,walk_buffers=5"
run slots8 "walk_run_slots=8,walk_buffers=8"
