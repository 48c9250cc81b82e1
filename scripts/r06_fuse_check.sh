#!/bin/bash
# Round 6: k_fused_rows (YKPRED_TUNE fuse_rows) on the two small-class populations:
# the parity tests of the zone-B writers, then the step's kernels with and without the decision branch.
export TMPDIR=/tmp
[ -n "$SKIP_TESTS" ] || timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "fused_rows or class_runs_writer or sweep_writer_equals or full_size_small_class" > gpurun_out/r06_fuse_tests.log 2>&1
tail -5 gpurun_out/r06_fuse_tests.log
TUNES=${1:-fuse_rows=0 fuse_combine=0 fuse_combine=1}
for T in $TUNES; do echo "== $T"; YKPRED_TUNE=$T timeout 300 python scripts/r06_own_alone.py 2>&1 | grep -v amdgpu.ids; YKPRED_TUNE=$T timeout 300 python scripts/r06_unique_alone.py 2>&1 | grep -v amdgpu.ids; done > gpurun_out/r06_fuse_own.txt 2>&1
cat gpurun_out/r06_fuse_own.txt
