#!/usr/bin/env python3
"""Developer tool: the last N kernel dispatches of a rocprofv3 rocpd database as a timeline
(start offset, duration, gap to the previous kernel's end).  prof_timeline.py <results.db> [N]"""
import sqlite3
import sys


def main():
    c = sqlite3.connect(sys.argv[1])
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 120
    views = [r[0] for r in c.execute("select name from sqlite_master where type in ('view','table')").fetchall()]
    v = "kernels" if "kernels" in views else [x for x in views if "kernel" in x and "dispatch" in x][0]
    cur = c.execute("select * from %s limit 1" % v)
    cols = [d[0] for d in cur.description]
    cur.fetchall()
    name = "name" if "name" in cols else [x for x in cols if "name" in x][0]
    start = "start" if "start" in cols else [x for x in cols if "start" in x][0]
    end = "end" if "end" in cols else [x for x in cols if "end" in x][0]
    rows = c.execute("select %s, %s, %s from %s order by %s" % (name, start, end, v, start)).fetchall()[-n:]
    t0, prev = rows[0][1], rows[0][1]
    for nm, s, e in rows:
        print("%10.2f us  dur %8.2f  gap %8.2f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev) / 1e3, nm.split("(")[0][-60:]))
        prev = e


if __name__ == "__main__":
    main()
