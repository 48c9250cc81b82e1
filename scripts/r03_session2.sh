#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
python -c "import importlib; importlib.import_module('yunikorn-k8shim_amd').build_all()" || exit 1
make -C oracle -s || exit 1
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
tail -25 gpurun_out/pytest_gpu.log
timeout 900 python bench.py --cpu-seconds 0 --steps 20 --warmup 5 > gpurun_out/bench_s2.json 2> gpurun_out/bench_s2.err
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/bench_s2.json") if l.startswith("{")][-1])
print("default ms", round(d["ms_per_step"], 4), "frac", d["roofline"]["whole_step_frac"], d["kernel_ms"])
for k, x in (d.get("variants") or {}).items():
    print("   ", k, x.get("ms_per_step"), (x.get("roofline") or {}).get("whole_step_frac"), x.get("kernel_ms"), x.get("error"), x.get("cold_pass"))
PY
timeout 900 python scripts/r03_shard_shapes.py 2>&1 | tail -20
