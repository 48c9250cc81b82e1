#!/bin/bash
# Round 5, final code: the whole GPU suite, the default bench line, the allocation rounds, then the rocprofv3 passes — kernel stats of
# the three ask populations and separate --pmc WRITE_SIZE / FETCH_SIZE passes of the same three (+ the calibration fill); configs[4]'s
# counters stay those of round 4 (traffic_r04.json: the band writer has not changed). Usage on the GPU box: bash scripts/r05_final.sh
O=gpurun_out/r05_final; mkdir -p $O
ROOT="$GRAFT_REPO_ROOT"; cd "$ROOT" || exit 1
timeout 1000 python -m pytest tests -x -q -m gpu --durations=6 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -9 $O/pytest_gpu.log
timeout 500 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -2 $O/bench.err
timeout 300 python scripts/bench_rounds.py --configs4 > $O/rounds.json 2> $O/rounds.err; echo "rounds rc=$?"
hipcc --offload-arch=gfx950 -O3 "$ROOT/scripts/pmc_calibration_fill.hip" -o /tmp/fill_probe 2>/dev/null
cd /tmp && export TMPDIR=/tmp
P="$ROOT/$O/pmc"; mkdir -p "$P"
COMMON="--steps 3 --warmup 1 --cpu-seconds 0 --profile-steps 0 --no-variants --no-ingest --no-verify"
for C in WRITE_SIZE FETCH_SIZE; do
  timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d "$P/default_$C" -- python "$ROOT/bench.py" $COMMON > "$P/default_$C.log" 2>&1
  timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d "$P/own_template_per_ask_$C" -- python "$ROOT/bench.py" $COMMON --templates 0 > "$P/own_$C.log" 2>&1
  timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d "$P/unique_request_vectors_$C" -- python "$ROOT/bench.py" $COMMON --templates 0 --unique-requests > "$P/unique_$C.log" 2>&1
  timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d "$P/fill_$C" -- /tmp/fill_probe > "$P/fill_$C.log" 2>&1
done
S="--cpu-seconds 0 --no-variants --no-ingest --no-verify"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$P/stats" -- python "$ROOT/bench.py" $S > "$P/stats_bench.log" 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$P/stats_own" -- python "$ROOT/bench.py" $S --templates 0 --steps 10 > "$P/stats_own.log" 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$P/stats_unique" -- python "$ROOT/bench.py" $S --templates 0 --unique-requests --steps 10 > "$P/stats_unique.log" 2>&1
# keep what is judged, drop the per-dispatch traces (the merge back is capped)
cd "$ROOT"
for d in stats stats_own stats_unique; do f=$(find $O/pmc/$d -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/kernel_stats_$d.csv; done
find $O/pmc -name "*kernel_trace.csv" -delete; find $O/pmc -name "*agent_info.csv" -delete
du -sh $O; find $O -name "*.csv" | head -20
