#!/bin/bash
# Round 6: batched rounds (parallel proposals, top-k lists, pair bits, node-by-node assume) — the round tests (one GPU and sharded over
# the stub), the prefix statistics, then rounds on ONE GPU in batches (YKPRED_TUNE round_batched=1) against the sequential kernel.
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_sequential.py -x -q -m gpu > gpurun_out/r06_batched_tests.log 2>&1
tail -8 gpurun_out/r06_batched_tests.log
bash scripts/r06_shard_round_stats.sh > gpurun_out/r06_shard_stats_after.txt 2>&1; grep round_prof gpurun_out/r06_shard_stats_after.txt
timeout 900 python scripts/r06_batched_one_gpu.py --configs4 --perf > gpurun_out/r06_batched_one_gpu.json 2> gpurun_out/r06_batched_one_gpu.err; grep round_prof gpurun_out/r06_batched_one_gpu.err | sort | uniq -c | tail -12; cat gpurun_out/r06_batched_one_gpu.json
