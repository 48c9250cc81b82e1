#!/bin/bash
# Round 6: a sweep of sharded allocation rounds over the stub transport (world 2..4 on one GPU): random cluster sizes, template counts,
# with and without hard spread constraints, different generator seeds — every rank's decisions against the oracle's sequential loop.
# (Every shard holds nodes: a rank with an empty shard has no evaluation to build a round on, and the ranks refuse together.)
# Usage on the GPU box: bash scripts/r06_shard_round_sweep.sh [cases]
ROOT="$GRAFT_REPO_ROOT"; cd "$ROOT"
/opt/rocm/bin/hipcc -O1 -fPIC -shared -std=c++17 tests/c/rccl_stub.cpp -o /tmp/librccl_stub.so -lrt || exit 1
N=${1:-24}; bad=0
for i in $(seq 1 $N); do
  W=$((2 + i % 3)); NODES=$((128 * W + (i * 97) % 900)); PODS=$((300 + (i * 131) % 1500)); T=$((1 + (i * 37) % 120)); S=$((i % 3 == 0 ? 1 : 0)); SEED=$((100 + i))
  out=$(SHARD_RCCL_STUB=/tmp/librccl_stub.so timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node=$W --master-addr 127.0.0.1 --master-port $((29600 + i)) tests/_shard_round_worker.py $NODES $PODS $T $S $SEED 2>&1)
  ok=$(echo "$out" | grep -o "sharded rounds True on_device True" | wc -l)
  echo "case $i: world $W nodes $NODES asks $PODS templates $T spread $S seed $SEED -> $ok of $W ranks equal the oracle: $(echo "$out" | grep -o "([0-9]* asks x [0-9]* nodes, [0-9]* allocated on [0-9]* nodes)" | head -1)"
  [ "$ok" = "$W" ] || { bad=$((bad + 1)); echo "$out" | tail -5; }
done
echo "sharded round sweep: $N cases, $bad bad"
