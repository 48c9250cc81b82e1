#!/bin/bash
# round 3, GPU session 15: the side benches with the final code (incremental paths, per-pair callback path, compressed gather at the
# shard shapes on one GPU, bench.py with two ranks on one GPU) + rocprofv3 kernel stats of the two small-class populations
cd "$(dirname "$0")/.." || exit 1
ROOT=$(pwd)
mkdir -p gpurun_out/r03_final
python -c "import importlib; importlib.import_module('yunikorn-k8shim_amd').build_all()" || exit 1
O=gpurun_out/r03_final
timeout 300 python scripts/bench_incremental.py > $O/r03_incremental.json 2> $O/incremental.err; echo "incremental rc=$?"
timeout 200 python scripts/bench_query.py > $O/r03_query.json 2> $O/query.err; echo "query rc=$?"
timeout 300 python scripts/bench_compressed_gather.py > $O/r03_compressed_gather_one_gpu.json 2> $O/gather.err; echo "gather rc=$?"
BENCH_BACKEND=gloo timeout 400 python bench.py --gpus 2 --steps 10 --warmup 3 --cpu-seconds 0 > $O/r03_bench_2ranks_one_gpu.json 2> $O/ranks2.err; echo "2 ranks rc=$?"
cd /tmp && export TMPDIR=/tmp
COMMON="--steps 10 --warmup 2 --cpu-seconds 0 --no-variants --no-ingest"
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/$O/stats_own" -- python "$ROOT/bench.py" $COMMON --templates 0 > "$ROOT/$O/stats_own.log" 2>&1; echo "stats own rc=$?"
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/$O/stats_unique" -- python "$ROOT/bench.py" $COMMON --templates 0 --unique-requests > "$ROOT/$O/stats_unique.log" 2>&1; echo "stats unique rc=$?"
cd "$ROOT"
rm -f $O/stats_*/*/*kernel_trace.csv $O/stats_*/*/*agent_info.csv
for f in $O/*.json; do echo "== $f"; head -c 700 $f; echo; done
for d in own unique; do echo "== stats $d"; head -8 $O/stats_$d/*/*kernel_stats.csv | cut -c1-160; done
