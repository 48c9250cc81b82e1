#!/bin/bash
# round 3, final GPU session: the whole GPU suite, then the bench line of the final code
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
python -c "import importlib; importlib.import_module('yunikorn-k8shim_amd').build_all()" || exit 1
make -C oracle -s || exit 1
timeout 1200 python -m pytest tests -m gpu -q --timeout 300 -p no:cacheprovider > gpurun_out/r03_pytest_gpu.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r03_pytest_gpu.log
grep -v "^\.\+ *\[" gpurun_out/r03_pytest_gpu.log | tail -12
timeout 600 python bench.py > gpurun_out/r03_bench_final.json 2> gpurun_out/r03_bench_final.err; echo "bench rc=$?"
python - <<'PY'
import json
d = json.load(open("gpurun_out/r03_bench_final.json"))
print(d["value"], d["ms_per_step"], d["roofline"])
for k, v in d.get("variants", {}).items():
    print(k, v.get("ms_per_step"), (v.get("roofline") or {}).get("whole_step_frac"), v.get("error"))
print(d.get("predicates_callback"))
print(d.get("end_to_end", {}).get("total_ms"), d.get("cpu_baseline", {}).get("value"))
PY
