#!/bin/bash
# Round 6: compile-time knobs of k_sweep_rows on the unique-request population.
# Usage on the GPU box: bash scripts/r06_sweep_builds.sh "<hipcc flags of one build>;<...>" "<YKPRED_TUNE> ..."
IFS=';' read -ra BUILDS <<< "${1:--DYK_SWEEP_BATCH=16}"
for V in "${BUILDS[@]}"; do
  YKPRED_EXTRA_HIPFLAGS="$V" python -c "
import importlib
b=importlib.import_module('yunikorn-k8shim_amd.build'); b.build_engine(force=True); b.build_host()" || exit 1
  for T in ${2:-"sweep_groups=0"}; do
    echo "== build [$V] tune=$T"
    YKPRED_TUNE="$T" python scripts/r06_unique_alone.py 2>&1 | grep -v amdgpu.ids | grep "decisions=False" | sed -e 's/k_planes+k_base_planes.*k_sig_planes/.../'
  done
done
