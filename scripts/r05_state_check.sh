#!/bin/bash
# Round 5: where the tree stands — whole GPU suite, the default bench line, the allocation rounds incl. the configs[4]-shaped one
O=gpurun_out/r05_state; mkdir -p $O
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 900 python -m pytest tests -x -q -m gpu --durations=8 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -12 $O/pytest_gpu.log
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -3 $O/bench.err
timeout 600 python scripts/bench_rounds.py --configs4 > $O/rounds.json 2> $O/rounds.err; echo "rounds rc=$?"; tail -3 $O/rounds.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r05_state/bench.json"))
print("ms", d["ms_per_step"], "verified", d.get("verified"), "roof", d["roofline"]["frac"], d["roofline"]["avg_launch_ms"], d["roofline"].get("whole_step_frac"))
print("kernel_ms", d.get("kernel_ms"))
for k, v in d.get("variants", {}).items():
    print(k, v.get("ms_per_step"), (v.get("roofline") or {}).get("whole_step_frac"), "verified", v.get("verified"), v.get("cold_pass"), v.get("error"))
    print("   ", v.get("kernel_ms"))
    if "allocation_round" in v: print("   round", v["allocation_round"])
print("rounds", json.dumps(d.get("allocation_round"), indent=1))
r = json.load(open("gpurun_out/r05_state/rounds.json"))
print("rounds-script", json.dumps(r, indent=1))
PY
