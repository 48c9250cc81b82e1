#!/bin/bash
# Round 4, GPU session 7: the self-verifying bench line (verify_leg on every leg, allocation_round leg) and configs[4] at its own size
mkdir -p gpurun_out/bench_and_configs4
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 1200 python bench.py > gpurun_out/bench_and_configs4/bench.json 2> gpurun_out/bench_and_configs4/bench.err; echo "bench rc=$?"
tail -3 gpurun_out/bench_and_configs4/bench.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/bench_and_configs4/bench.json"))
print("value", d["value"], "ms", d["ms_per_step"], "verified", d.get("verified"), d.get("verification"))
print("roofline", d["roofline"])
for k, v in d.get("variants", {}).items():
    print(k, v.get("ms_per_step"), (v.get("roofline") or {}).get("whole_step_frac"), "verified", v.get("verified"), v.get("verification"), v.get("error"))
print("rounds", json.dumps(d.get("allocation_round"), indent=1))
print(d.get("predicates_callback"))
PY
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "configs4-size" -s > gpurun_out/bench_and_configs4/pytest_configs4_size.log 2>&1
echo "configs4-size rc=$?"; tail -5 gpurun_out/bench_and_configs4/pytest_configs4_size.log
