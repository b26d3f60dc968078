#!/bin/bash
# quick FETCH_SIZE / WRITE_SIZE per dispatch of the heavy kernels with the library as built (run on the GPU box):
#   DEPTH=1 bash tools/experiments/fetch_quick.sh       (KB per dispatch as the counter reports them: x2 on gfx950, MI355X_MICROARCH.md)
export TMPDIR=/tmp
R=$PWD
BENCH="python $R/bench.py --steps 8 --warmup 4 --windows 2 --depth ${DEPTH:-1} --no-cpu-baseline --no-secondary --no-e2e --no-sustained"
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pq_$c
  (cd /tmp && timeout 300 rocprofv3 --pmc $c --kernel-trace -d /tmp/pq_$c -o pmc -- $BENCH > /tmp/pq_$c.log 2>&1 < /dev/null)
  python3 - /tmp/pq_$c/pmc_results.db <<'PY'
import sys, sqlite3, re
cur = sqlite3.connect(sys.argv[1]).cursor()
cols = [d[1] for d in cur.execute("pragma table_info('counters_collection')")]
namecol = "counter_name" if "counter_name" in cols else "name"
kcol = "kernel_name" if "kernel_name" in cols else "name"
for k, c, v, n in cur.execute(f"select {kcol}, {namecol}, avg(value), count(*) from counters_collection group by {kcol}, {namecol}"):
    k = re.sub(r"\(.*", "", k).replace("void ", "")
    if "k_vit" in k or "front_stream<" in k:
        print(c, k, n, round(v, 1))
PY
done
