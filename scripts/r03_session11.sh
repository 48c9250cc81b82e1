#!/bin/bash
# round 3, GPU session 11: where k_combine_slices' time goes — mode bits (timing only) and SQ / TCC counters of the unique-request population
cd "$(dirname "$0")/.." || exit 1
ROOT=$(pwd)
mkdir -p gpurun_out
rm -f gpurun_out/r03_probe2.jsonl
rm -rf gpurun_out/pmc_slices
python -c "import importlib; importlib.import_module('yunikorn-k8shim_amd').build_all()" || exit 1
if [ -z "$PMC_ONLY" ]; then
export PROBE_SETS='[
 {"knobs":{"YKPRED_SLICE_PAIRS":"1"},"workloads":"unique","both":true},
 {"knobs":{"YKPRED_SLICE_PAIRS":"1","YKPRED_SLICE_MODE":"5"},"workloads":"unique","both":true},
 {"knobs":{"YKPRED_SLICE_PAIRS":"1","YKPRED_SLICE_MODE":"13"},"workloads":"unique","both":true},
 {"knobs":{"YKPRED_SLICE_PAIRS":"1","YKPRED_SLICE_MODE":"29"},"workloads":"unique","both":true},
 {"knobs":{"YKPRED_SLICE_PAIRS":"1","YKPRED_SLICE_MODE":"16"},"workloads":"unique","both":true}
]'
timeout 600 python scripts/r03_probe2.py 2>&1 | grep '^{' | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l)
    print(d['knobs'], d['workload'], d['ms_per_step'], d.get('ms_per_step_nodec'), 'k_combine', d['kernel_ms'].get('k_combine'), 'alone', d.get('kernel_ms_nodec', {}).get('k_combine'))"
fi
mkdir -p "$ROOT/gpurun_out/pmc_slices"
cd /tmp && export TMPDIR=/tmp
export PROBE_SETS='[{"knobs":{"YKPRED_SLICE_PAIRS":"1"},"workloads":"unique"}]'
export PROBE_OUT=r03_probe2_pmc.jsonl
P1="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU"
P2="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS"
P3="FETCH_SIZE"
P4="WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"
i=0
for P in "$P1" "$P2" "$P3" "$P4"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $P --kernel-trace --output-format csv -d "$ROOT/gpurun_out/pmc_slices/p$i" -- python "$ROOT/scripts/r03_probe2.py" > "$ROOT/gpurun_out/pmc_slices/p$i.log" 2>&1
  echo "pass $i rc=$?"
done
cd "$ROOT"
python - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob("gpurun_out/pmc_slices/p*/*/*counter_collection.csv")):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("ykk::", "").replace("void ", "")
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    print("==", f.split("/")[2])
    for k, cs in agg.items():
        if any(t in k for t in ("k_combine", "k_dim_walk", "k_decide", "k_sig_planes")):
            print("  ", k[:40], {c: round(sum(v) / len(v)) for c, v in cs.items()}, "launches", len(next(iter(cs.values()))))
PY
rm -rf gpurun_out/pmc_slices/p*/*/*agent_info.csv
