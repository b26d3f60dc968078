#!/bin/bash
# batches in flight x hardware queues of the runtime (GPU_MAX_HW_QUEUES): rate, step time, window spread
F="--steps 20 --warmup 8 --no-cpu-baseline --no-secondary --no-e2e --no-sustained"
one() { python bench.py $F "$@" 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); t=d['timing']
print('q=$GPU_MAX_HW_QUEUES $*', round(d['value']/1e9,3), round(d['ms_per_step'],4), 'spread', round(t['window_spread'],3), 'all', round(t['all_windows_ms_per_step'],4), 'hostcpu', round(list(d['breakdown_ms'].values())[0],3))"; }
for q in 8 12 16 24; do
export GPU_MAX_HW_QUEUES=$q
for d in 8 12 16 24; do one --depth $d; done
done
