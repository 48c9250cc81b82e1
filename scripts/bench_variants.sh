#!/bin/bash
# Ask-population variants of bench.py with engine tunables swept (run on the GPU box): prints ms/step and k_combine per run.
out=${1:-gpurun_out/r02/variants}
mkdir -p "$out"
run() {  # name, env..., -- args
  name=$1; shift
  envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" python bench.py --cpu-seconds 0 --no-variants --steps 10 --warmup 2 "$@" > "$out/$name.json" 2> "$out/$name.err"
  python - "$out/$name.json" "$name" <<PY
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    k = d["kernel_ms"]
    print(f"{sys.argv[2]:40s} {d['ms_per_step']:9.3f} ms  combine {k.get('k_combine')}  classes {d['config']['pod_classes']} planes {d['config']['signature_planes']}  {dict((a, b) for a, b in k.items() if b > 0.3)}")
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
run default A=1 --
run own_tpl_block YKPRED_WAVE_COMBINE_BELOW=-1 -- --templates 0
run own_tpl_wave YKPRED_WAVE_COMBINE_BELOW=64 -- --templates 0
run unique_block YKPRED_WAVE_COMBINE_BELOW=-1 -- --templates 0 --unique-requests
run unique_wave A=1 -- --templates 0 --unique-requests
run unique_block_nocap YKPRED_WAVE_COMBINE_BELOW=-1 YKPRED_COMBINE_LDS=-1 -- --templates 0 --unique-requests
run gang100 A=1 -- --gang 100
run templates20k A=1 -- --templates 20000
run templates20k_wave YKPRED_WAVE_COMBINE_BELOW=1000 -- --templates 20000
