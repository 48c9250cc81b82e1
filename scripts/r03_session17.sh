#!/bin/bash
# round 3, GPU session 17: fuzz sweeps of the final code (bitmap, counts, per-pair kernel, failing plugin, decisions vs the oracle)
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
python -c "import importlib; importlib.import_module('yunikorn-k8shim_amd').build_all()" || exit 1
make -C oracle -s || exit 1
L=gpurun_out/r03_fuzz_parity.log
: > $L
( echo "# default tunables"; timeout 300 python scripts/fuzz_parity.py 300000 160 ) >> $L 2>&1
( echo "# YKPRED_WALK_ROWS=1"; YKPRED_WALK_ROWS=1 timeout 300 python scripts/fuzz_parity.py 310000 160 ) >> $L 2>&1
( echo "# YKPRED_WALK_ROWS=2 YKPRED_SIG_WPL=4"; YKPRED_WALK_ROWS=2 YKPRED_SIG_WPL=4 timeout 300 python scripts/fuzz_parity.py 320000 120 ) >> $L 2>&1
( echo "# YKPRED_WALK_ROWS=1 YKPRED_COMBINE_SLICES=0 YKPRED_SIG_WPL=2"; YKPRED_WALK_ROWS=1 YKPRED_COMBINE_SLICES=0 YKPRED_SIG_WPL=2 timeout 300 python scripts/fuzz_parity.py 330000 80 ) >> $L 2>&1
( echo "# incremental"; timeout 300 python scripts/fuzz_incremental.py 340000 40 20 ) >> gpurun_out/r03_fuzz_incremental.log 2>&1
cat $L | tail -20; tail -3 gpurun_out/r03_fuzz_incremental.log
