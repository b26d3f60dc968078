#!/bin/bash
# does bringing up a process group change the decode-only rate?  (round 4: --force-gather's decode-only phase ran 10 % faster than the plain run)
F="--steps 20 --warmup 8 --depth 4 --no-cpu-baseline --no-secondary --no-e2e"
one() { python bench.py $F "$@" 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); x=d.get('decode_only', None)
if x is None: x=dict(value=d['value'], ms_per_step=d['ms_per_step'], window_spread=d['timing']['window_spread'])
print('$BENCH_DUMMY_STREAMS $*', round(x['value']/1e9,3), round(x['ms_per_step'],4), 'spread', round(x['window_spread'],3))"; }
for n in 0 1 2 3 5; do
export BENCH_DUMMY_STREAMS=$n
one
one
done
export BENCH_DUMMY_STREAMS=0
one --force-gather
