#!/bin/bash
# same-box A/B of bench.py under environment variants: tools/ab_env.sh <tag> <workload> "<ENV1=.. ENV2=..>" ["<variant 2>" ...]
tag=$1; wl=$2; shift 2
mkdir -p gpurun_out/$tag
i=0
for envs in "$@"; do
  env $envs python bench.py --workload $wl --steps ${STEPS:-40} --warmup 8 --no-cpu-baseline --no-roofline 2>gpurun_out/$tag/${wl}_v$i.err | tail -1 > gpurun_out/$tag/${wl}_v$i.json
  python -c "
import json;d=json.load(open('gpurun_out/$tag/${wl}_v$i.json'));print('$wl [$envs]', round(d['ms_per_step'],3), round(d['value'],1))"
  i=$((i+1))
done
