#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd database (kernel-trace [+ pmc]) as text: per-kernel calls,
total/mean duration, share.  Usage: prof_summary.py <results.db> [out.txt]"""
import sqlite3
import sys


def main():
    db = sys.argv[1]
    out = open(sys.argv[2], "w") if len(sys.argv) > 2 else sys.stdout
    c = sqlite3.connect(db)
    rows = c.execute("select name, total_calls, total_duration, average, percentage from top_kernels").fetchall()
    out.write("# rocprofv3 --kernel-trace --stats summary (durations in us)\n")
    out.write("%-72s %8s %14s %10s %7s\n" % ("kernel", "calls", "total_us", "avg_us", "pct"))
    for name, calls, total, avg, pct in rows[:15]:
        short = name.split("(")[0][-70:]
        out.write("%-72s %8d %14.1f %10.3f %7.2f\n" % (short, calls, total, avg, pct))
    try:
        pm = c.execute("select * from pmc_events limit 1").fetchall()
        if pm:
            cur = c.execute("select * from counters_collection limit 0")
            cols = [d[0] for d in cur.description]
            out.write("\n# counters_collection columns: %s\n" % cols)
    except Exception as e:  # noqa
        pass
    if out is not sys.stdout:
        out.close()


if __name__ == "__main__":
    main()
