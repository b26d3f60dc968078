set -x
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r05_gputest_a.log 2>&1; echo "pytest rc=$?"
tail -5 gpurun_out/r05_gputest_a.log
timeout 600 python bench.py > gpurun_out/r05_bench_a.json 2> gpurun_out/r05_bench_a.err; echo "bench rc=$?"
tail -c 600 gpurun_out/r05_bench_a.err
timeout 900 bash tools/prof_run.sh r05d8 mix 8; echo "prof rc=$?"
head -40 gpurun_out/prof_r05d8/summary.md
