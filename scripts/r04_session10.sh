#!/bin/bash
# Round 4, GPU session 10: SQ counters of k_walk_rows on the unique-request population
mkdir -p gpurun_out/r04s10
ROOT="$GRAFT_REPO_ROOT"
cd /tmp && export TMPDIR=/tmp
COMMON="--templates 0 --unique-requests --steps 2 --warmup 1 --cpu-seconds 0 --profile-steps 0 --no-variants --no-ingest --no-verify"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d "$ROOT/gpurun_out/r04s10/p1" -- python "$ROOT/bench.py" $COMMON > "$ROOT/gpurun_out/r04s10/p1.log" 2>&1
rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVES SQ_INSTS_SMEM --kernel-trace --output-format csv -d "$ROOT/gpurun_out/r04s10/p2" -- python "$ROOT/bench.py" $COMMON > "$ROOT/gpurun_out/r04s10/p2.log" 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_WAIT_INST_ANY SQ_LDS_ADDR_CONFLICT --kernel-trace --output-format csv -d "$ROOT/gpurun_out/r04s10/p3" -- python "$ROOT/bench.py" $COMMON > "$ROOT/gpurun_out/r04s10/p3.log" 2>&1
cd "$ROOT"
python - <<'PY'
import csv, glob, collections
for p in ("p1", "p2", "p3"):
    files = glob.glob(f"gpurun_out/r04s10/{p}/*/*counter_collection.csv")
    if not files:
        print(p, "no counter file"); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(files[0])):
        k = r["Kernel_Name"].split("(")[0]
        if "k_walk_rows" in k or "k_expand_bands" in k:
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in agg.items():
        print(p, k[:30], {c: round(sum(v) / len(v)) for c, v in cs.items()}, "launches", len(next(iter(cs.values()))))
PY
tail -3 gpurun_out/r04s10/p3.log
