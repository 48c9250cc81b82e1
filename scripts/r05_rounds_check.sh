#!/bin/bash
# Round 5: the allocation-round kernel on its own — the sequential-parity tests, then the rounds bench with the per-phase ticks
O=gpurun_out/r05_rounds; mkdir -p $O
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 400 python -m pytest tests/test_gpu_sequential.py -x -q > $O/pytest_seq.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest_seq.log
YKPRED_TUNE=round_prof=1 timeout 300 python scripts/bench_rounds.py --configs4 > $O/rounds.json 2> $O/rounds.err; echo "rounds rc=$?"
grep round_prof $O/rounds.err
python - <<'PY'
import json
r = json.load(open("gpurun_out/r05_rounds/rounds.json"))
for k, v in r.items():
    if isinstance(v, dict): print(k, v.get("asks"), "alloc/s", round(v.get("allocations_per_sec", 0)), "ms", v.get("round_ms"), "us/ask", v.get("us_per_ask"), "verified", v.get("verified"), "on_device", v.get("on_device", v.get("stats")))
PY
