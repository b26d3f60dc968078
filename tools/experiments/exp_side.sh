#!/bin/bash
# the plans' side streams off (default) / on (--side-stream, the round-3 form) at several numbers of batches in flight
F="--steps 20 --warmup 8 --no-cpu-baseline --no-secondary --no-e2e --no-sustained"
one() { python bench.py $F "$@" 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); x=d.get('decode_only', None)
if x is None: x=dict(value=d['value'], ms_per_step=d['ms_per_step'], window_spread=d['timing']['window_spread'], host_cpu_ms_per_step=list(d['breakdown_ms'].values())[0])
print('$*', round(x['value']/1e9,3), round(x['ms_per_step'],4), 'spread', round(x['window_spread'],3), 'hostcpu', round(x['host_cpu_ms_per_step'],3))"; }
for ns in "" "--side-stream"; do
one $ns --depth 4
one $ns --depth 4
one $ns --depth 8
one $ns --depth 3
one $ns --depth 5
one $ns --depth 6
done
