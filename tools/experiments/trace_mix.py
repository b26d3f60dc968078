#!/usr/bin/env python3
"""What runs beside what: from a rocprofv3 kernel trace (rocpd sqlite) of the pipelined bench, the share of the steady-state
time in which the GPU runs (a) nothing, (b) only light kernels (walks, lists, copies: a few workgroups), (c) n heavy kernels
(front end / trellis) at once, and which heavy kinds overlap.  usage: trace_mix.py <trace_results.db>"""
import collections
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1]).cursor()
rows = c.execute("select name,start,end,queue_id from kernels order by start").fetchall()
lo = rows[len(rows) // 3][1]
hi = rows[-len(rows) // 6][2]


def kind(n):
    n = n.split("(")[0].replace("void ", "")
    if n.startswith("k_front_stream") and "fix" not in n:
        return "F"
    if n.startswith("k_vit<1") or n.startswith("k_vit<2"):
        return "T"
    return "l"


ev = []
for n, s, e, q in rows:
    if e > lo and s < hi:
        ev += [(max(s, lo), 1, kind(n)), (min(e, hi), -1, kind(n))]
ev.sort()
cnt = collections.Counter()
state = collections.Counter()
last = lo
for t, d, k in ev:
    key = "F%d T%d l%d" % (state["F"], state["T"], min(state["l"], 1))
    cnt[key] += t - last
    state[k] += d
    last = t
tot = hi - lo
print("steady-state window %.1f ms" % (tot / 1e6))
for k, v in sorted(cnt.items(), key=lambda x: -x[1])[:14]:
    print("%-12s %5.1f %%" % (k, 100.0 * v / tot))
idle = sum(v for k, v in cnt.items() if k.startswith("F0 T0"))
print("no heavy kernel running: %.1f %%" % (100.0 * idle / tot))
