#!/bin/bash
# Round 4, GPU session 18: the GPU suite with the new k_walk_rows (the guard-subset re-runs and the 70 s configs[4]-size oracle leg were run before: r04_pytest_gpu_session11.log)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/s18
timeout 700 python -m pytest tests -q -m gpu -k "not subset_under_the_guard and not configs4-size" -p no:cacheprovider > gpurun_out/s18/pytest_gpu.log 2>&1
tail -6 gpurun_out/s18/pytest_gpu.log
