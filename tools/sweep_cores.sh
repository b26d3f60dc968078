# the default bench confined to C cores (what a rank gets on a box with few cores per GPU), W = C host threads
for c in 1 2 4; do
taskset -c 0-$((c-1)) python bench.py --steps 60 --warmup 16 --no-cpu-baseline --no-secondary --sync-threads $c 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); b=d['breakdown_ms']; print('cores', $c, 'threads', $c, round(d['ms_per_step'],3), 'ms/step', round(d['value']/1e9,3), 'e9 bursts/s; cpu ms/step', round(list(b.values())[0],3))"
done
