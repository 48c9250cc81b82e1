#!/bin/bash
# PMC passes for the roofline `traffic` field: WRITE_SIZE and FETCH_SIZE in SEPARATE rocprofv3 runs (TCC slot limits),
# on bench.py (configs[2]) and on the fill probe (calibration: that kernel writes exactly 6 272 000 000 bytes).
# Usage on the GPU box: bash scripts/pmc_passes.sh <outdir>
set -u
OUT=${1:-gpurun_out/pmc}
mkdir -p "$OUT"
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
hipcc --offload-arch=gfx950 -O3 "$ROOT/scripts/fill_probe.hip" -o /tmp/fill_probe 2>/dev/null
cd /tmp && export TMPDIR=/tmp
for C in WRITE_SIZE FETCH_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d "$ROOT/$OUT/bench_$C" -- python "$ROOT/bench.py" --steps 3 --warmup 1 --cpu-seconds 0 --profile-steps 0 --no-variants > "$ROOT/$OUT/bench_$C.log" 2>&1
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d "$ROOT/$OUT/fill_$C" -- /tmp/fill_probe > "$ROOT/$OUT/fill_$C.log" 2>&1
done
find "$ROOT/$OUT" -name "*.csv" | head -20
