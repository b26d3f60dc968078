#!/usr/bin/env python3
"""Summarise rocprofv3 (ROCm 7.2, rocpd sqlite output) runs made by tools/prof_run.sh:
   kernel-trace stats per kernel, and per-kernel averages of every collected PMC counter."""
import glob
import os
import re
import sqlite3
import sys


def short(n):
    n = re.sub(r"\(.*", "", n)
    return n.replace("void ", "")


def trace_stats(db):
    cur = sqlite3.connect(db).cursor()
    rows = cur.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) "
                       "from kernels group by name order by sum(duration) desc").fetchall()
    tot = sum(r[2] for r in rows) or 1
    out = ["| kernel | calls | total us | avg us | min us | max us | % |", "|---|---|---|---|---|---|---|"]
    for name, calls, total, avg, mn, mx in rows:
        out.append(f"| {short(name)} | {calls} | {total/1e3:.1f} | {avg/1e3:.2f} | {mn/1e3:.2f} | {mx/1e3:.2f} | {100*total/tot:.1f} |")
    return "\n".join(out)


def pmc_stats(db):
    cur = sqlite3.connect(db).cursor()
    cols = [d[1] for d in cur.execute("pragma table_info('counters_collection')")]
    namecol = "counter_name" if "counter_name" in cols else "name"
    kcol = "kernel_name" if "kernel_name" in cols else "name"
    return cols, cur.execute(f"select {kcol}, {namecol}, avg(value), count(*) from counters_collection "
                             f"group by {kcol}, {namecol} order by {kcol}, {namecol}").fetchall()


def main():
    d = sys.argv[1]
    print(f"# rocprofv3 summary of {os.path.basename(d.rstrip('/'))}\n")
    c = os.path.join(d, "cmd.txt")
    if os.path.exists(c):
        print("command (every pass: `rocprofv3 <pass options> -- <this>`): `%s`\n" % open(c).read().strip())
    t = os.path.join(d, "trace", "trace_results.db")
    if os.path.exists(t):
        print("## kernel trace (rocprofv3 --kernel-trace --stats, command in the title line)\n")
        print(trace_stats(t))
        print()
    for db in sorted(glob.glob(os.path.join(d, "pmc_*", "pmc_results.db"))):
        print(f"## PMC pass {os.path.basename(os.path.dirname(db))} (average per dispatch)\n")
        try:
            cols, rows = pmc_stats(db)
        except Exception as e:
            print("could not read:", e)
            continue
        print("| kernel | counter | avg per dispatch | dispatches |")
        print("|---|---|---|---|")
        for k, c, v, n in rows:
            print(f"| {short(k)} | {c} | {v:.6g} | {n} |")
        print()


if __name__ == "__main__":
    main()
