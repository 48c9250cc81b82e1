#!/bin/bash
# round 3, GPU session 19: memory-path counters of the slice writer (SQ VMEM levels / FIFO stalls, TA, TCC, EA stalls), unique-request population
cd "$(dirname "$0")/.." || exit 1
ROOT=$(pwd)
python -c "import importlib; importlib.import_module('yunikorn-k8shim_amd').build_all()" || exit 1
rm -rf gpurun_out/pmc_slices2; mkdir -p gpurun_out/pmc_slices2
cd /tmp && export TMPDIR=/tmp
export PROBE_SETS='[{"knobs":{},"workloads":"unique"}]'
export PROBE_OUT=r03_probe2_pmc.jsonl
P1="SQ_WAVES SQ_LEVEL_WAVES SQ_INST_LEVEL_VMEM SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
P2="TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_TA_BUSY_sum TA_TOTAL_WAVEFRONTS_sum"
P3="TCC_BUSY_avr TCC_CYCLE_sum TCC_ATOMIC_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum"
P4="TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_TAG_STALL_sum TCC_EA0_WRREQ_sum"
P5="SQ_BUSY_CU_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU GRBM_GUI_ACTIVE"
i=0
for P in "$P1" "$P2" "$P3" "$P4" "$P5"; do
  i=$((i+1))
  timeout 120 rocprofv3 --pmc $P --kernel-trace --output-format csv -d "$ROOT/gpurun_out/pmc_slices2/p$i" -- python "$ROOT/scripts/r03_probe2.py" > "$ROOT/gpurun_out/pmc_slices2/p$i.log" 2>&1
  echo "pass $i rc=$?"
done
cd "$ROOT"
python - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob("gpurun_out/pmc_slices2/p*/*/*counter_collection.csv")):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("ykk::", "").replace("void ", "")
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    print("==", f.split("/")[2])
    for k, cs in agg.items():
        if any(t in k for t in ("k_combine_slices", "k_expand", "k_dim_walk")):
            print("  ", k[:40], {c: round(sum(v) / len(v)) for c, v in cs.items()}, "launches", len(next(iter(cs.values()))))
PY
rm -f gpurun_out/pmc_slices2/p*/*/*agent_info.csv gpurun_out/pmc_slices2/p*/*/*kernel_trace.csv
grep -l "rror" gpurun_out/pmc_slices2/p*.log | head; grep -h -i "invalid\|not found\|unknown" gpurun_out/pmc_slices2/p*.log | head -5
