#!/bin/bash
# Round 6: candidates per ask (kPropK) of the batched rounds: 4 against 8 on the configs[2] round of one GPU.
for K in 8 4; do
  YKPRED_EXTRA_HIPFLAGS="-DYK_PROP_K=$K" python -c "
import importlib
b=importlib.import_module('yunikorn-k8shim_amd.build'); b.build_engine(force=True); b.build_host()" || exit 1
  echo "== kPropK $K"
  timeout 600 python scripts/r06_batched_one_gpu.py 2>&1 | grep -E "round_prof batched.*20000 asks|allocations_per_sec|equal" | sort | uniq -c
done
