#!/bin/bash
# Round 4, GPU session 13: PMC passes (three populations + configs[4] whole on one GPU + calibration fill) and kernel stats
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/pmc_r04
bash scripts/pmc_passes.sh gpurun_out/pmc_r04 > gpurun_out/pmc_r04/passes.log 2>&1
python scripts/summarize_pmc.py gpurun_out/pmc_r04 r04 > gpurun_out/pmc_r04/summary.txt 2>&1
cp profiles/r04_pmc_summary.json profiles/traffic_r04.json gpurun_out/pmc_r04/ 2>/dev/null
tail -40 gpurun_out/pmc_r04/summary.txt
for d in stats stats_unique stats_configs4; do
  f=$(ls gpurun_out/pmc_r04/$d/*/*kernel_stats.csv 2>/dev/null | head -1)
  echo "== $d: $f"; head -8 "$f"
done
find gpurun_out/pmc_r04 -name "*counter_collection.csv" -size +20M -delete
find gpurun_out/pmc_r04 -name "*kernel_trace.csv" -size +5M -delete
du -sh gpurun_out/pmc_r04
