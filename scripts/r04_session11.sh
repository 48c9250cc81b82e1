#!/bin/bash
# Round 4, GPU session 11: failure injection, the C-ABI collectives with world > 1 through the stub, the pruned engine (whole suite)
mkdir -p gpurun_out/r04s11
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "failure_injection or c_abi_collectives" > gpurun_out/r04s11/pytest_new.log 2>&1
echo "new tests rc=$?"; tail -15 gpurun_out/r04s11/pytest_new.log
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/r04s11/pytest_gpu.log 2>&1
echo "suite rc=$?"; tail -4 gpurun_out/r04s11/pytest_gpu.log
