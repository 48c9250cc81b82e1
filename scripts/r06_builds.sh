#!/bin/bash
# Round 6: compile-time experiment switches of a kernel against one of the population scripts.
# Usage on the GPU box: bash scripts/r06_builds.sh "<hipcc flags of one build>;<...>" <script> ["<YKPRED_TUNE> ..."]
IFS=';' read -ra BUILDS <<< "${1:- }"
for V in "${BUILDS[@]}"; do
  YKPRED_EXTRA_HIPFLAGS="$V" python -c "
import importlib
b=importlib.import_module('yunikorn-k8shim_amd.build'); b.build_engine(force=True); b.build_host()" || exit 1
  for T in ${3:-"class_runs=1"}; do
    echo "== build [$V] tune=$T"
    YKPRED_TUNE="$T" YKPRED_TRACE_RUNS=1 python $2 2>&1 | grep -v amdgpu.ids | grep -E "decisions=False|^runs:" | sort -u | sed -e 's/k_planes+k_base_planes.*k_class_rows/.../'
  done
done
