#!/bin/bash
# PMC passes for the roofline `traffic` fields: WRITE_SIZE and FETCH_SIZE in SEPARATE rocprofv3 runs (TCC slot limits; --pmc is
# never combined with the hip/hsa/memory-copy trace domains), on bench.py's three ask populations — configs[2] default (2 000
# templates), every ask its own template, a distinct cpu request per ask — and on the fill probe (calibration: that kernel
# writes exactly 6 272 000 000 bytes). Plus one `--kernel-trace --stats` run of the default bench for the per-kernel durations.
# Usage on the GPU box: bash scripts/pmc_passes.sh <outdir>
set -u
OUT=${1:-gpurun_out/pmc}
mkdir -p "$OUT"
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
hipcc --offload-arch=gfx950 -O3 "$ROOT/scripts/pmc_calibration_fill.hip" -o /tmp/fill_probe 2>/dev/null
cd /tmp && export TMPDIR=/tmp
COMMON="--steps 3 --warmup 1 --cpu-seconds 0 --profile-steps 0 --no-variants --no-ingest --no-verify"
for C in WRITE_SIZE FETCH_SIZE; do
  timeout 600 rocprofv3 --pmc $C --kernel-trace --output-format csv -d "$ROOT/$OUT/default_$C" -- python "$ROOT/bench.py" $COMMON > "$ROOT/$OUT/default_$C.log" 2>&1
  timeout 600 rocprofv3 --pmc $C --kernel-trace --output-format csv -d "$ROOT/$OUT/own_template_per_ask_$C" -- python "$ROOT/bench.py" $COMMON --templates 0 > "$ROOT/$OUT/own_$C.log" 2>&1
  timeout 600 rocprofv3 --pmc $C --kernel-trace --output-format csv -d "$ROOT/$OUT/unique_request_vectors_$C" -- python "$ROOT/bench.py" $COMMON --templates 0 --unique-requests > "$ROOT/$OUT/unique_$C.log" 2>&1
  timeout 600 rocprofv3 --pmc $C --kernel-trace --output-format csv -d "$ROOT/$OUT/fill_$C" -- /tmp/fill_probe > "$ROOT/$OUT/fill_$C.log" 2>&1
  # BASELINE configs[4] whole on one GPU: 100 000 nodes x 5 000 000 asks, hard spread constraints on 10 % of the templates
  timeout 600 rocprofv3 --pmc $C --kernel-trace --output-format csv -d "$ROOT/$OUT/configs4_one_gpu_$C" -- python "$ROOT/bench.py" $COMMON --nodes 100000 --pods 5000000 --spread --seed-offset 2 > "$ROOT/$OUT/configs4_$C.log" 2>&1
done
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/$OUT/stats" -- python "$ROOT/bench.py" --cpu-seconds 0 --no-variants --no-ingest --no-verify > "$ROOT/$OUT/stats_bench.log" 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/$OUT/stats_unique" -- python "$ROOT/bench.py" --cpu-seconds 0 --no-variants --no-ingest --no-verify --templates 0 --unique-requests --steps 10 > "$ROOT/$OUT/stats_unique.log" 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/$OUT/stats_configs4" -- python "$ROOT/bench.py" --cpu-seconds 0 --no-variants --no-ingest --no-verify --nodes 100000 --pods 5000000 --spread --seed-offset 2 --steps 5 --warmup 1 > "$ROOT/$OUT/stats_configs4.log" 2>&1
find "$ROOT/$OUT" -name "*.csv" | head -30
