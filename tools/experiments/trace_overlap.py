#!/usr/bin/env python3
"""Timeline statistics of a rocprofv3 kernel trace (rocpd sqlite): busy fraction, mean concurrency, per-kernel time
in the steady-state half of the run.  usage: trace_overlap.py <trace_results.db>"""
import collections
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1]).cursor()
rows = c.execute("select name,start,end from kernels order by start").fetchall()
lo = rows[len(rows) // 2][1]
hi = rows[-1][2]
ev = []
per = collections.defaultdict(lambda: [0, 0.0])
for n, s, e in rows:
    if s >= lo:
        ev += [(s, 1), (e, -1)]
        k = n.split("(")[0].replace("void ", "")
        per[k][0] += 1
        per[k][1] += (e - s) / 1e3
ev.sort()
busy = area = conc = 0
last = lo
for t, d in ev:
    if conc > 0:
        busy += t - last
    area += conc * (t - last)
    conc += d
    last = t
w = (hi - lo) / 1e3
print(f"window {w:.0f} us, GPU busy {busy / 1e3 / w:.3f}, mean concurrency while busy {area / max(busy, 1):.2f}, "
      f"sum of kernel durations / window {area / 1e3 / w:.3f}")
steps = per.get("k_front_stream", [1])[0]
for k, (n, t) in sorted(per.items(), key=lambda x: -x[1][1]):
    print(f"{k:36s} calls {n:5d}  avg {t / n:8.1f} us   per step {t / steps:8.1f} us")
