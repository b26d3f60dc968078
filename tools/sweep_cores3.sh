for i in 1 2 3; do
taskset -c 0-1 python bench.py --steps 80 --warmup 16 --no-cpu-baseline --no-secondary 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('policy', round(d['ms_per_step'],3), d['config']['host_threads_per_gpu'])"
taskset -c 0-1 python bench.py --steps 80 --warmup 16 --no-cpu-baseline --no-secondary --sync-threads 2 --blocking-sync 0 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('W=2 spin', round(d['ms_per_step'],3))"
done
