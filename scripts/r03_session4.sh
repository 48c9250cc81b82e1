#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
rm -f gpurun_out/r03_variant_probe.jsonl
python -c "import importlib; importlib.import_module('yunikorn-k8shim_amd').build_all()" || exit 1
make -C oracle -s || exit 1
timeout 900 python -m pytest tests -m gpu -q --timeout 240 -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
grep -v "^\.\+ *\[" gpurun_out/pytest_gpu.log | tail -40
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_s4.json 2> gpurun_out/bench_s4.err
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/bench_s4.json") if l.startswith("{")][-1])
print("default ms", round(d["ms_per_step"], 4), "frac", d["roofline"], d["kernel_ms"])
for k, x in (d.get("variants") or {}).items():
    print("   ", k, x.get("ms_per_step"), (x.get("roofline") or {}).get("whole_step_frac"), x.get("kernel_ms"), x.get("error"))
print("e2e", d.get("end_to_end"))
print("cpu", d.get("cpu_baseline"))
PY
probe() { env "$@" timeout 200 python scripts/r03_variant_probe.py 2>&1 | grep '^{' | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['knobs'], d['workload'], d['ms_per_step'], 'S', d['band_steps'], 'band_rows', d['band_rows'], d['kernel_ms'])"; }
probe YKPRED_ZONE_B_FIRST=1
probe YKPRED_COMBINE_WORDS=1
probe YKPRED_COMBINE_WORDS=2
probe YKPRED_PERMUTE_ALL=0
