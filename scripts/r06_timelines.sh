#!/bin/bash
# Round 6: the kernel timeline (scripts/r06_timeline.py) of one step of the two small-class populations. Usage on the GPU box:
# bash scripts/r06_timelines.sh [tag]
T=${1:-now}; ROOT="$GRAFT_REPO_ROOT"; O="$ROOT/gpurun_out/r06_tl_$T"; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
S="--cpu-seconds 0 --no-variants --no-ingest --no-verify --profile-steps 0 --steps 6 --warmup 2"
timeout 300 rocprofv3 --kernel-trace --output-format csv -d "$O/own" -- python "$ROOT/bench.py" $S --templates 0 > "$O/own.log" 2>&1
timeout 300 rocprofv3 --kernel-trace --output-format csv -d "$O/unique" -- python "$ROOT/bench.py" $S --templates 0 --unique-requests > "$O/unique.log" 2>&1
cd "$ROOT"
for w in own unique; do f=$(find $O/$w -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && python scripts/r06_timeline.py "$f" > $O/timeline_$w.txt; done
find $O -name "*.csv" -delete
cat $O/timeline_own.txt $O/timeline_unique.txt
