#!/bin/bash
# A/B of an environment knob on one box: tools/ab_env.sh VAR v1 v2 ... (two bench runs per value: rate, step time, trellis kernels)
VAR=$1; shift
for v in "$@"; do
  for i in 1 2; do
    env $VAR=$v python bench.py --steps 20 --warmup 8 --depth 4 --no-cpu-baseline --no-secondary --no-e2e 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); k=list(d['breakdown_ms'].values())[-1]
print('[$VAR=$v]', round(d['value']/1e9,3), round(d['ms_per_step'],4), 'front', round(k['k_front_stream']*1e3,1), 'vit216', round(k['k_vit<216>']*1e3,1), 'vit432', round(k['k_vit<432>']*1e3,1))"
  done
done
