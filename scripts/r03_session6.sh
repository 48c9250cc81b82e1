#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
rm -f gpurun_out/r03_variant_probe.jsonl
python -c "import importlib; importlib.import_module('yunikorn-k8shim_amd').build_all()" || exit 1
make -C oracle -s || exit 1
timeout 900 python -m pytest tests -m gpu -q --timeout 240 -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
grep -v "^\.\+ *\[" gpurun_out/pytest_gpu.log | tail -40
probe() { env "$@" timeout 200 python scripts/r03_variant_probe.py 2>&1 | grep '^{' | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['knobs'], d['workload'], d['ms_per_step'], d['kernel_ms'])"; }
probe YKPRED_DECIDE_SKIP=1
probe YKPRED_DECIDE_SKIP=0
