#!/bin/bash
# the default bench at several pipeline depths (batches in flight) / hardware-queue counts: rate, step time, windows, spread
# usage: tools/spread.sh "<depth>:<GPU_MAX_HW_QUEUES>" ...
for spec in "$@"; do d=${spec%%:*}; q=${spec#*:}; for i in 1 2; do GPU_MAX_HW_QUEUES=$q python bench.py --depth $d --warmup $((3*d)) --no-cpu-baseline --no-secondary --no-e2e --no-sustained 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); t=d['timing']
print('depth $d queues $q', round(d['value']/1e9,3), round(d['ms_per_step'],4), [round(x,4) for x in t['windows_ms_per_step']], round(t['window_spread'],3), 'cpu', round(list(d['breakdown_ms'].values())[0],3))"; done; done
