#!/bin/bash
# Round 6: what ends the conflict-free prefixes of a sharded allocation round (world 2 on one GPU over the stub transport).
# Usage on the GPU box: bash scripts/r06_shard_round_stats.sh [nodes pods templates spread]
ROOT="$GRAFT_REPO_ROOT"; cd "$ROOT"
/opt/rocm/bin/hipcc -O1 -fPIC -shared -std=c++17 tests/c/rccl_stub.cpp -o /tmp/librccl_stub.so -lrt || exit 1
run() { W=$1; shift; SHARD_RCCL_STUB=/tmp/librccl_stub.so YKPRED_TUNE=round_prof=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node=$W --master-addr 127.0.0.1 --master-port 29917 tests/_shard_round_worker.py "$@" 2>&1 | grep -E "round_prof batched|sharded rounds" | sort | uniq -c; }
if [ $# -ge 3 ]; then run 2 "$@"; else
  echo "== world 2, 5120 nodes, 3000 asks, 300 templates"; run 2 5120 3000 300 0
  echo "== world 2, 5120 nodes, 3000 asks, 300 templates, hard spread on a tenth"; run 2 5120 3000 300 1
  echo "== world 3, 5120 nodes, 3000 asks, 2000 templates"; run 3 5120 3000 2000 0
fi
