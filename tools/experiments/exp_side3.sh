#!/bin/bash
F="--steps 20 --warmup 8 --no-cpu-baseline --no-secondary --no-e2e --no-sustained"

one() { python bench.py $F "$@" 2>gpurun_out/err.txt | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); x=d.get('decode_only', None)
if x is None: x=dict(value=d['value'], ms_per_step=d['ms_per_step'], window_spread=d['timing']['window_spread'], host_cpu_ms_per_step=list(d['breakdown_ms'].values())[0], w=d['timing']['windows_ms_per_step'])
print('q=$GPU_MAX_HW_QUEUES $*', round(x['value']/1e9,3), round(x['ms_per_step'],4), 'spread', round(x['window_spread'],3), x.get('w'))"; grep "step completion" gpurun_out/err.txt | head -1 | cut -c1-400; }
one --depth 8
one --depth 8
one --depth 4
export GPU_MAX_HW_QUEUES=8
one --depth 8
one --depth 8
one --depth 4
export GPU_MAX_HW_QUEUES=6
one --depth 6
one --depth 12
unset GPU_MAX_HW_QUEUES
one --depth 12
one --depth 16
