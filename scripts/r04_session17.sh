#!/bin/bash
# Round 4, GPU session 17: index-row paths with the whole-row k_walk_rows — parity subset, plain and under the exact-end guard
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/s17
K="sorted_walk or slice_writer or unique_request or failure_injection or incremental"
timeout 400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_sequential.py -x -q -m gpu -k "$K" > gpurun_out/s17/pytest_plain.log 2>&1
tail -4 gpurun_out/s17/pytest_plain.log
YKPRED_GUARD_PAGES=1 timeout 400 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "sorted_walk or slice_writer or unique_request" > gpurun_out/s17/pytest_guard1.log 2>&1
tail -4 gpurun_out/s17/pytest_guard1.log
