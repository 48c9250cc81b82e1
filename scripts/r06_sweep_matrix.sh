#!/bin/bash
# Round 6: the unique-request population (bench.py --templates 0 --unique-requests) over the sweep writer's two knobs.
# Usage on the GPU box: bash scripts/r06_sweep_matrix.sh "<min_run values>" "<group values>"
O=gpurun_out/r06_sweep; mkdir -p $O
for M in ${1:-2}; do for G in ${2:-0}; do
  YKPRED_TUNE="sweep_min_run=$M,sweep_groups=$G" timeout 300 python bench.py --steps 10 --cpu-seconds 0 --no-variants --no-ingest --templates 0 --unique-requests > $O/u_m${M}_g$G.json 2> $O/u_m${M}_g$G.err || tail -3 $O/u_m${M}_g$G.err
  python - <<PY
import json
d=json.load(open("$O/u_m${M}_g$G.json")); k=d.get("kernel_ms",{})
print("min_run=$M groups=$G ms=%.3f verified=%s"%(d["ms_per_step"], d.get("verified")), {n:k[n] for n in k if n in("k_dim_walk","k_sig_planes","k_sweep_rows","k_slice_desc","k_walk_rows","k_combine","k_dim_walk(ranked)","k_decide")})
PY
done; done
