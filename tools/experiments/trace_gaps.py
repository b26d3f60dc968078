#!/usr/bin/env python3
"""Per kernel of the pipelined bench, from a rocprofv3 kernel trace (rocpd sqlite): average duration and the average time
between the end of the kernel in front of it on the same queue and its own start (what it waited for a place on the chip or
for the packet processor), over the steady-state part of the run.  usage: trace_gaps.py <trace_results.db>"""
import collections
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1]).cursor()
cols = [r[1] for r in c.execute("pragma table_info(kernels)").fetchall()]
key = "stream_id" if "stream_id" in cols else "queue_id"
rows = c.execute("select name,start,end,%s from kernels order by start" % key).fetchall()
lo = rows[len(rows) // 3][1]
hi = rows[-len(rows) // 6][2]
last = {}
dur = collections.defaultdict(list)
gap = collections.defaultdict(list)
for n, s, e, q in rows:
    n = n.split("(")[0].replace("void ", "")
    if s >= lo and e <= hi:
        dur[n].append(e - s)
        if q in last:
            gap[n].append(s - last[q])
    last[q] = e
print("by %s; steady-state window %.1f ms" % (key, (hi - lo) / 1e6))
print("%-28s %6s %9s %9s %9s" % ("kernel", "calls", "avg us", "gap us", "gap max"))
tg = td = 0.0
for n in sorted(dur, key=lambda x: -sum(dur[x])):
    g = gap.get(n, [0])
    print("%-28s %6d %9.1f %9.1f %9.1f" % (n[:28], len(dur[n]), sum(dur[n]) / len(dur[n]) / 1e3, sum(g) / max(1, len(g)) / 1e3, max(g) / 1e3))
    td += sum(dur[n]) / len(dur[n]) / 1e3
    tg += sum(g) / max(1, len(g)) / 1e3
print("sum of averages: kernels %.1f us, gaps %.1f us" % (td, tg))
