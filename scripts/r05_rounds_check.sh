#!/bin/bash
# Round 5: the allocation-round kernel on its own — the sequential-parity tests, the rounds bench, and (when a -DYK_ROUND_PROF build of
# the engine lies next to the product library) the same rounds again with the per-phase ticks of the loop
O=gpurun_out/r05_rounds; mkdir -p $O
cd "$GRAFT_REPO_ROOT" || exit 1
L=yunikorn-k8shim_amd/lib
if [ "${1:-}" != "prof-only" ]; then
timeout 400 python -m pytest tests/test_gpu_sequential.py -x -q > $O/pytest_seq.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest_seq.log
timeout 300 python scripts/bench_rounds.py --configs4 > $O/rounds.json 2> $O/rounds.err; echo "rounds rc=$?"
python - <<'PY'
import json
r = json.load(open("gpurun_out/r05_rounds/rounds.json"))
for k, v in r.items():
    if isinstance(v, dict): print(k, v.get("asks"), "alloc/s", round(v.get("allocations_per_sec", 0)), "ms", v.get("round_ms"), "us/ask", v.get("us_per_ask"), "verified", v.get("verified"), "on_device", v.get("on_device", v.get("stats")))
PY
fi
if [ -f $L/libykpred_prof.so ]; then
  cp $L/libykpred.so /tmp/libykpred_product.so; cp $L/libykpred_prof.so $L/libykpred.so
  YKPRED_TUNE=round_prof=1 timeout 300 python scripts/bench_rounds.py --configs4 > $O/rounds_prof.json 2> $O/rounds_prof.err; echo "prof rounds rc=$?"
  grep round_prof $O/rounds_prof.err
  cp /tmp/libykpred_product.so $L/libykpred.so
fi
