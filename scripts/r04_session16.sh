#!/bin/bash
# Round 4, GPU session 16: kernel trace of the unique-request workload (grid, LDS, registers and duration of every k_walk_rows launch)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/s16
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d gpurun_out/s16/trace -o run -- python bench.py --unique-requests --steps 6 --warmup 2 --no-variants --cpu-seconds 0 --no-verify --profile-steps 0 > gpurun_out/s16/bench.json 2> gpurun_out/s16/bench.err
f=$(ls gpurun_out/s16/trace/*/*kernel_trace.csv | head -1)
python - "$f" <<'PY'
import csv, sys, collections
rows=list(csv.DictReader(open(sys.argv[1])))
print(list(rows[0].keys()))
agg=collections.defaultdict(list)
for r in rows:
    n=r['Kernel_Name']
    if 'k_walk_rows' in n or 'k_slice_desc' in n or 'k_combine' in n:
        key=(n[:60], r.get('Grid_Size_X'), r.get('Grid_Size_Y'), r.get('Workgroup_Size_X'), r.get('LDS_Block_Size'), r.get('VGPR_Count'), r.get('Accum_VGPR_Count'), r.get('SGPR_Count'), r.get('Scratch_Size'))
        agg[key].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
for k,v in agg.items():
    print(k, 'n=%d avg=%.1f us min=%.1f'%(len(v), sum(v)/len(v), min(v)))
PY
head -12 $(ls gpurun_out/s16/trace/*/*kernel_stats.csv | head -1)
find gpurun_out/s16 -name "*kernel_trace.csv" -size +5M -delete
