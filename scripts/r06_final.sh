#!/bin/bash
# Round 6, final code: the whole GPU suite, the default bench line, then the rocprofv3 passes — kernel stats of the four ask
# populations (configs[2], own template, unique requests, configs[4] whole) and separate --pmc WRITE_SIZE / FETCH_SIZE /
# SQ_INSTS_VALU passes of the same four (+ the calibration fill). Usage on the GPU box: bash scripts/r06_final.sh [skip-suite]
O=gpurun_out/r06_final; mkdir -p $O
ROOT="$GRAFT_REPO_ROOT"; cd "$ROOT" || exit 1
if [ "${1:-}" != "skip-suite" ]; then
  timeout 1200 python -m pytest tests -x -q -m gpu --durations=6 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -9 $O/pytest_gpu.log
fi
timeout 700 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -2 $O/bench.err
hipcc --offload-arch=gfx950 -O3 "$ROOT/scripts/pmc_calibration_fill.hip" -o /tmp/fill_probe 2>/dev/null
cd /tmp && export TMPDIR=/tmp
P="$ROOT/$O/pmc"; mkdir -p "$P"
COMMON="--steps 3 --warmup 1 --cpu-seconds 0 --profile-steps 0 --no-variants --no-ingest --no-verify"
C4="--nodes 100000 --pods 5000000 --spread --seed-offset 2"
for C in WRITE_SIZE FETCH_SIZE SQ_INSTS_VALU; do
  timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d "$P/default_$C" -- python "$ROOT/bench.py" $COMMON > "$P/default_$C.log" 2>&1
  timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d "$P/own_template_per_ask_$C" -- python "$ROOT/bench.py" $COMMON --templates 0 > "$P/own_$C.log" 2>&1
  timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d "$P/unique_request_vectors_$C" -- python "$ROOT/bench.py" $COMMON --templates 0 --unique-requests > "$P/unique_$C.log" 2>&1
  timeout 400 rocprofv3 --pmc $C --kernel-trace --output-format csv -d "$P/configs4_one_gpu_$C" -- python "$ROOT/bench.py" $COMMON $C4 > "$P/configs4_$C.log" 2>&1
  [ "$C" != "SQ_INSTS_VALU" ] && timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d "$P/fill_$C" -- /tmp/fill_probe > "$P/fill_$C.log" 2>&1
done
# int-ops/eval of the per-pair ablation kernel (k_direct: one compare chain per pair), one step
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU --kernel-trace --output-format csv -d "$P/direct_SQ_INSTS_VALU" -- python "$ROOT/bench.py" --steps 1 --warmup 0 --cpu-seconds 0 --profile-steps 0 --no-variants --no-ingest --no-verify --direct > "$P/direct_valu.log" 2>&1
S="--cpu-seconds 0 --no-variants --no-ingest --no-verify"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$P/stats" -- python "$ROOT/bench.py" $S > "$P/stats_bench.log" 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$P/stats_own" -- python "$ROOT/bench.py" $S --templates 0 --steps 10 > "$P/stats_own.log" 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$P/stats_unique" -- python "$ROOT/bench.py" $S --templates 0 --unique-requests --steps 10 > "$P/stats_unique.log" 2>&1
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$P/stats_configs4" -- python "$ROOT/bench.py" $S $C4 --steps 5 --warmup 1 > "$P/stats_configs4.log" 2>&1
# keep what is judged, drop the per-dispatch traces (the merge back is capped)
cd "$ROOT"
for d in stats stats_own stats_unique stats_configs4; do f=$(find $O/pmc/$d -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/kernel_stats_$d.csv; done
python scripts/summarize_pmc.py $O/pmc r06 | tail -30
f=$(find $O/pmc/direct_SQ_INSTS_VALU -name "*counter_collection.csv" | head -1); [ -n "$f" ] && grep k_direct "$f" | head -3 > $O/direct_valu.txt
find $O/pmc -name "*kernel_trace.csv" -delete; find $O/pmc -name "*agent_info.csv" -delete; find $O/pmc -name "*counter_collection.csv" -size +2M -delete
du -sh $O; ls $O
