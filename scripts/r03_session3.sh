#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
python -c "import importlib; importlib.import_module('yunikorn-k8shim_amd').build_all()" || exit 1
make -C oracle -s || exit 1
timeout 1000 python -m pytest tests -m gpu -v --timeout 240 -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
grep -v "PASSED" gpurun_out/pytest_gpu.log | tail -60
timeout 400 python scripts/r03_variant_probe.py 2>&1 | tail -8
