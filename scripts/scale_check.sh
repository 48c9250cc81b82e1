#!/bin/bash
# scale_check.sh — the FIRST run on a multi-GPU node is a measurement with its own parity assertion, not a bring-up.
# For world = 2, 4, 8 (as many as the node has GPUs), one process per GPU, real RCCL over xGMI through the C ABI
# (ykpred_comm_init / the histogram all-reduce inside ykpred_eval / ykpred_gather_bitmap[_compressed] / ykpred_exchange_decisions):
#   1. tests/_shard_worker.py: every rank compares EVERY gathered row (plain and class-compressed gather), every exchanged feasible
#      count and every exchanged decision with one engine that holds the whole cluster (hard spread constraints: the histograms
#      are summed across shards); a rank that finds a difference exits 3 and the curve is not printed for that world size;
#   1b. tests/_shard_round_worker.py: two allocation rounds decided by the ranks together (proposals per batch, one all-gather each)
#      against the oracle's sequential loop over the whole cluster;
#   2. bench.py --gpus W (BASELINE configs[3]: 50 000 nodes sharded W-way x 1M gang-placeholder asks, gathered bitmap in the step) and
#      bench.py --gpus W --no-gather (decisions + counts only: 16 bytes per ask over the links) — each verifies its own shard slabs.
# Output: one JSON line per (world, mode) under $OUT (default gpurun_out/scale), and the curve on stdout. Needs >= 2 GPUs.
set -u
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
OUT="${1:-$ROOT/gpurun_out/scale}"
mkdir -p "$OUT"
cd "$ROOT"
export HSA_ENABLE_IPC_MODE_LEGACY=0
GPUS=$(python - <<'PY'
import torch
print(torch.cuda.device_count())
PY
)
if [ "$GPUS" -lt 2 ]; then echo "scale_check: $GPUS GPU(s) visible, nothing to scale over"; exit 2; fi
python bench.py --gpus 1 --no-variants --cpu-seconds 0 > "$OUT/n1.json" 2> "$OUT/n1.err" || { echo "world 1 bench failed"; tail -5 "$OUT/n1.err"; exit 1; }
RC=0
for W in 2 4 8; do
  [ "$W" -le "$GPUS" ] || continue
  PORT=$((29500 + W))
  echo "== world $W: parity of the collectives against one engine (50 048 nodes x 200 000 asks, hard spread constraints)"
  if ! timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node "$W" --master-addr 127.0.0.1 --master-port "$PORT" \
        tests/_shard_worker.py 50048 200000 > "$OUT/check_w$W.log" 2>&1; then
    echo "   FAILED — see $OUT/check_w$W.log"; tail -8 "$OUT/check_w$W.log"; RC=1; continue
  fi
  grep "^rank" "$OUT/check_w$W.log"
  echo "== world $W: allocation rounds on the sharded cluster against the oracle's sequential loop (12 800 nodes x 6 000 asks)"
  if ! timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node "$W" --master-addr 127.0.0.1 --master-port "$((PORT + 20))" \
        tests/_shard_round_worker.py 12800 6000 200 > "$OUT/rounds_w$W.log" 2>&1; then
    echo "   FAILED — see $OUT/rounds_w$W.log"; tail -8 "$OUT/rounds_w$W.log"; RC=1
  else
    grep "^rank" "$OUT/rounds_w$W.log"
  fi
  timeout 900 python bench.py --gpus "$W" --cpu-seconds 0 > "$OUT/n${W}_gathered.json" 2> "$OUT/n${W}_gathered.err" || { echo "   gathered bench failed"; RC=1; }
  timeout 900 python bench.py --gpus "$W" --no-gather --cpu-seconds 0 > "$OUT/n${W}_decisions_only.json" 2> "$OUT/n${W}_decisions_only.err" || { echo "   decisions-only bench failed"; RC=1; }
done
python - "$OUT" <<'PY'
import glob, json, os, sys
out = sys.argv[1]
rows = []
for f in sorted(glob.glob(os.path.join(out, "n*.json"))):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as exc:  # noqa: BLE001
        print(f"{os.path.basename(f)}: unreadable ({exc})")
        continue
    rows.append((d["n_gpus"], os.path.basename(f), d["ms_per_step"], d["value"], d.get("verified"), d["config"].get("collectives")))
base = next((r for r in rows if r[0] == 1), None)
print(f"{'gpus':>4} {'line':<28} {'ms/step':>9} {'evals/s':>12} {'vs 1 GPU':>9}  verified  collectives")
for n, name, ms, val, ok, coll in sorted(rows):
    rel = f"{val / base[3]:.2f}x" if base else "-"
    print(f"{n:>4} {name:<28} {ms:>9.3f} {val:>12.4g} {rel:>9}  {ok}  {coll}")
PY
exit $RC
