#!/usr/bin/env python3
"""What runs beside what in the pipelined bench WITHOUT a profiler (a -DTG_TRACE build: the heavy kernels' workgroups stamp
their first and last tick of the 100 MHz clock, bench.py --trace-dump).  From the stamps of the steady-state part: the share of
the time in which no / one / two / ... kinds of heavy kernel have a workgroup running, the average number of workgroups of each
kind in flight, and each launch's duration (first workgroup's start to last workgroup's end).
usage: trace_untraced.py dump.npy [every n-th front-end workgroup stamped: 4 / TG_STREAM_WPB]"""
import collections
import sys

import numpy as np

a = np.load(sys.argv[1])
kind = (a[:, 0] & 0xFFFFFFFF).astype(np.int64)
t0 = a[:, 1].astype(np.int64)
t1 = a[:, 2].astype(np.int64)
lo = np.sort(t0)[len(t0) // 3]
hi = np.sort(t1)[-len(t1) // 6]
names = {0: "front", 1: "sb1", 2: "vit216", 3: "vit432"}
weight = {0: int(sys.argv[2]) if len(sys.argv) > 2 else 1, 1: 8, 2: 8, 3: 8}        # every n-th workgroup of the trellis kernels is stamped
m = (t1 > lo) & (t0 < hi)
ev = []
for k, s, e in zip(kind[m], np.maximum(t0[m], lo), np.minimum(t1[m], hi)):
    ev.append((s, 1, k))
    ev.append((e, -1, k))
ev.sort()
state = collections.Counter()
share = collections.Counter()
busy = collections.Counter()
last = lo
for t, d, k in ev:
    key = "+".join(names[q] for q in sorted(state) if state[q] > 0) or "none"
    share[key] += t - last
    for q in state:
        busy[q] += state[q] * (t - last)
    state[k] += d
    last = t
tot = hi - lo
print("steady-state window %.2f ms (100 MHz ticks)" % (tot / 1e5))
for k, v in sorted(share.items(), key=lambda x: -x[1]):
    print("  %-24s %5.1f %%" % (k, 100.0 * v / tot))
for q in sorted(busy):
    print("  workgroups of %-7s in flight on average: %.0f" % (names[q], weight[q] * busy[q] / tot))
# launches: gaps of more than 20 us between the sorted starts of one kind separate them
for q in (0, 2, 3):
    s = np.sort(t0[(kind == q) & m])
    cut = np.flatnonzero(np.diff(s) > 2000)
    print("  %-7s: %d launches in the window" % (names[q], len(cut) + 1))
