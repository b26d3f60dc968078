#!/bin/bash
F="--steps 20 --warmup 8 --depth 4 --no-cpu-baseline --no-secondary --no-e2e"
one() { python bench.py $F "$@" 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); x=d.get('decode_only', None)
if x is None: x=dict(value=d['value'], ms_per_step=d['ms_per_step'], window_spread=d['timing']['window_spread'], host_cpu_ms_per_step=list(d['breakdown_ms'].values())[0])
print('comm=$BENCH_INIT_COMM $*', round(x['value']/1e9,3), round(x['ms_per_step'],4), 'spread', round(x['window_spread'],3), 'hostcpu', round(x['host_cpu_ms_per_step'],3))"; }
export BENCH_INIT_COMM=1
one
one
unset BENCH_INIT_COMM
(while true; do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|Power" | tr "\n" " "; echo; sleep 1; done) > gpurun_out/smi_plain.txt &
SM=$!
one
kill $SM
(while true; do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|Power" | tr "\n" " "; echo; sleep 1; done) > gpurun_out/smi_nccl.txt &
SM=$!
one --force-gather --wire-form grid
kill $SM
tail -4 gpurun_out/smi_plain.txt; echo; tail -4 gpurun_out/smi_nccl.txt
bash tools/prof_quick.sh plain $F | head -14
bash tools/prof_quick.sh nccl $F --force-gather --wire-form grid | head -16
