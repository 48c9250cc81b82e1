#!/bin/bash
# Round 4, GPU session 8: k_walk_rows (loader / store waves) — parity first, then the unique-request population
mkdir -p gpurun_out/r04s8
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "slice_writer or sorted_walk or unique_request or incremental_fuzz or sub_wave" > gpurun_out/r04s8/pytest_walk.log 2>&1
echo "pytest walk rc=$?"; tail -5 gpurun_out/r04s8/pytest_walk.log
env YKPRED_TUNE=walk_rows=1 timeout 600 python scripts/fuzz_parity.py 370000 40 > gpurun_out/r04s8/fuzz_parity_walk1.log 2>&1; echo "fuzz_parity walk1: $(tail -1 gpurun_out/r04s8/fuzz_parity_walk1.log)"
env YKPRED_TUNE=walk_rows=1 YKPRED_GUARD_PAGES=1 timeout 600 python scripts/fuzz_incremental.py 371000 16 12 > gpurun_out/r04s8/fuzz_incr_walk1_guard.log 2>&1; echo "fuzz_incr walk1 guard: $(tail -1 gpurun_out/r04s8/fuzz_incr_walk1_guard.log)"
timeout 900 python bench.py --templates 0 --unique-requests --no-variants --cpu-seconds 0 --steps 10 > gpurun_out/r04s8/bench_unique.json 2> gpurun_out/r04s8/bench_unique.err; echo "bench unique rc=$?"
python - <<'PY'
import json
d = json.load(open("gpurun_out/r04s8/bench_unique.json"))
print("unique: ms", d["ms_per_step"], "verified", d.get("verified"), "kernels", d["kernel_ms"], "whole", d["roofline"]["whole_step_frac"])
PY
