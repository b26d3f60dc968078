for cfg in "2 2 0" "2 3 0" "2 4 0" "2 3 1" "2 4 1" "2 6 1" "1 2 1" "1 3 1"; do set -- $cfg
taskset -c 0-$(($1-1)) python bench.py --steps 60 --warmup 16 --no-cpu-baseline --no-secondary --sync-threads $2 --blocking-sync $3 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); b=d['breakdown_ms']; print('cores', $1, 'threads', $2, 'blocking', $3, round(d['ms_per_step'],3), 'ms/step; cpu ms/step', round(list(b.values())[0],3))"
done
