#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd database as text.
  kernel-trace:  per-kernel calls / total / mean duration / share   (view top_kernels)
  --pmc passes:  per-kernel mean of every collected counter         (view counters_collection)
Usage: prof_summary.py <results.db> [out.txt]"""
import sqlite3
import sys


def main():
    db = sys.argv[1]
    out = open(sys.argv[2], "w") if len(sys.argv) > 2 else sys.stdout
    c = sqlite3.connect(db)
    try:
        rows = c.execute("select name, total_calls, total_duration, average, percentage from top_kernels").fetchall()
    except Exception:
        rows = []
    if rows:
        out.write("# rocprofv3 --kernel-trace --stats summary (durations in us)\n")
        out.write("%-72s %8s %14s %10s %7s\n" % ("kernel", "calls", "total_us", "avg_us", "pct"))
        for name, calls, total, avg, pct in rows[:15]:
            short = name.split("(")[0][-70:]
            out.write("%-72s %8d %14.1f %10.3f %7.2f\n" % (short, calls, total, avg, pct))
    try:
        cur = c.execute("select * from counters_collection limit 1")
        cols = [d[0] for d in cur.description]
        if cur.fetchall():
            kcol = "kernel_name" if "kernel_name" in cols else [x for x in cols if "name" in x][0]
            ccol = "counter_name" if "counter_name" in cols else [x for x in cols if "counter" in x and "name" in x][0]
            vcol = "value" if "value" in cols else [x for x in cols if "value" in x][0]
            q = "select %s, %s, count(*), avg(%s), sum(%s) from counters_collection group by %s, %s order by sum(%s) desc" % (
                kcol, ccol, vcol, vcol, kcol, ccol, vcol)
            out.write("\n# PMC counters per kernel (mean per dispatch)\n")
            out.write("%-60s %-16s %8s %16s\n" % ("kernel", "counter", "calls", "mean"))
            for name, cn, n, av, sm in c.execute(q).fetchall()[:30]:
                out.write("%-60s %-16s %8d %16.1f\n" % (name.split("(")[0][-58:], cn, n, av))
    except Exception as e:  # noqa
        out.write("# (no counters: %s)\n" % e)
    if out is not sys.stdout:
        out.close()


if __name__ == "__main__":
    main()
