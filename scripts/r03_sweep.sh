#!/bin/bash
# Round-3 GPU session 1: the GPU suite, then bench.py under the writer knobs (zone-B beside / after the band writer; band height).
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
python -c "import importlib; importlib.import_module('yunikorn-k8shim_amd').build_all()" || exit 1
make -C oracle -s || exit 1
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
tail -5 gpurun_out/pytest_gpu.log
run() { # name, env...
  name=$1; shift
  env "$@" timeout 600 python bench.py --cpu-seconds 0 --steps 20 --warmup 5 > gpurun_out/bench_$name.json 2> gpurun_out/bench_$name.err
  python - "$name" <<'PY'
import json, sys
name = sys.argv[1]
try:
    d = json.loads([l for l in open(f"gpurun_out/bench_{name}.json") if l.startswith("{")][-1])
    v = d.get("variants") or {}
    print(name, "default ms", round(d["ms_per_step"], 4), "frac", d["roofline"]["whole_step_frac"], {k: round(x, 3) for k, x in d["kernel_ms"].items()})
    for k, x in v.items():
        print("   ", k, x.get("ms_per_step"), (x.get("roofline") or {}).get("whole_step_frac"), x.get("kernel_ms"), x.get("error"))
except Exception as e:
    print(name, "FAILED", e)
PY
}
run auto
run serial YKPRED_COMBINE_SERIAL=1
run s128 YKPRED_BAND_STEPS=128
run s128_serial YKPRED_BAND_STEPS=128 YKPRED_COMBINE_SERIAL=1
run s64 YKPRED_BAND_STEPS=64
run s32 YKPRED_BAND_STEPS=32
