#!/bin/bash
# Round 4, GPU session 19: PMC passes + kernel stats of the unique-request workload again (k_walk_rows was rewritten after session 13)
cd "$GRAFT_REPO_ROOT" || exit 1
ROOT=$GRAFT_REPO_ROOT; OUT=gpurun_out/pmc_r04b; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
COMMON="--steps 3 --warmup 1 --cpu-seconds 0 --profile-steps 0 --no-variants --no-ingest --no-verify"
for C in WRITE_SIZE FETCH_SIZE; do
  timeout 150 rocprofv3 --pmc $C --kernel-trace --output-format csv -d "$ROOT/$OUT/unique_request_vectors_$C" -- python "$ROOT/bench.py" $COMMON --templates 0 --unique-requests > "$ROOT/$OUT/unique_$C.log" 2>&1
done
timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/$OUT/stats_unique" -- python "$ROOT/bench.py" --cpu-seconds 0 --no-variants --no-ingest --no-verify --templates 0 --unique-requests --steps 10 > "$ROOT/$OUT/stats_unique.log" 2>&1
cd $ROOT
python - <<'PY'
import csv, glob, json, collections
def per_kernel(counter):
    f=glob.glob(f"gpurun_out/pmc_r04b/unique_request_vectors_{counter}/*/*counter_collection.csv")[0]
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"]==counter: agg[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
    return {k:{"launches":len(v),"avg_KiB":sum(v)/len(v),"min_KiB":min(v),"max_KiB":max(v)} for k,v in agg.items()}
out={"WRITE_SIZE":per_kernel("WRITE_SIZE"),"FETCH_SIZE":per_kernel("FETCH_SIZE")}
json.dump(out,open("gpurun_out/pmc_r04b/unique_per_kernel.json","w"),indent=1)
for k,v in out["WRITE_SIZE"].items():
    if "walk_rows" in k or "slice_desc" in k or "dim_" in k: print(k, v, out["FETCH_SIZE"].get(k))
PY
head -8 $(ls $OUT/stats_unique/*/*kernel_stats.csv | head -1)
find $OUT -name "*counter_collection.csv" -size +20M -delete; find $OUT -name "*kernel_trace.csv" -size +5M -delete
