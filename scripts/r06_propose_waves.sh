#!/bin/bash
# Round 6: waves per workgroup of k_round_propose (a workgroup per ask): the kernel's average under rocprofv3 and the round's time.
for WV in 4 8 16; do
  YKPRED_EXTRA_HIPFLAGS="-DYK_PROPOSE_WAVES=$WV" python -c "
import importlib
b=importlib.import_module('yunikorn-k8shim_amd.build'); b.build_engine(force=True); b.build_host()" 2>&1 | grep -iE "error|spill" | head -3
  echo "== $WV waves"
  timeout 600 python scripts/r06_batched_one_gpu.py 2>&1 | grep -E "\"allocations_per_sec\"|equal|20000 asks" | sort | uniq -c | cut -c1-60,330-420
  (cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -- python $GRAFT_REPO_ROOT/scripts/r06_batched_one_gpu.py > /dev/null 2>&1; f=$(find /tmp/prof -name "*kernel_stats.csv" | head -1); grep -E "k_round" $f | cut -d, -f1-4 | cut -c1-120)
done
