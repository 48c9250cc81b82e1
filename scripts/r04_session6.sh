#!/bin/bash
mkdir -p gpurun_out/r04s6
cd "$GRAFT_REPO_ROOT" || exit 1
for i in 1 2 3; do
env YKPRED_GUARD_PAGES=3 timeout 300 python scripts/fuzz_parity.py 360000 30 > gpurun_out/r04s6/poison_parity_$i.log 2>&1
echo "poison parity run $i: $(tail -1 gpurun_out/r04s6/poison_parity_$i.log)"
done
timeout 900 python -m pytest tests/test_gpu_sequential.py -x -q -m gpu > gpurun_out/r04s6/pytest_seq.log 2>&1
echo "pytest_seq rc=$?"; tail -30 gpurun_out/r04s6/pytest_seq.log
