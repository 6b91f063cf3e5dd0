#!/usr/bin/env python3
"""Summarises rocprofv3 rocpd (.db) outputs: per-kernel time stats and per-kernel PMC averages.
usage: rocpd_summary.py <dir-with-*.db> [...]"""
import glob
import os
import sqlite3
import sys


def cols(c, t):
    return [r[1] for r in c.execute("pragma table_info(%s)" % t)]


def summarize(db):
    c = sqlite3.connect(db)
    out = []
    try:
        kc = cols(c, "kernels")
        name_col = "name" if "name" in kc else [k for k in kc if "name" in k][0]
        rows = c.execute("select %s, count(*), avg(end-start), min(end-start), max(end-start), sum(end-start) from kernels "
                         "group by %s order by sum(end-start) desc" % (name_col, name_col)).fetchall()
        tot = sum(r[5] for r in rows) or 1
        out.append("KERNEL_STATS (ns): name | calls | avg | min | max | total | pct")
        for r in rows[:12]:
            out.append("  %-60s | %5d | %10.0f | %8d | %8d | %12d | %5.1f%%" % (r[0][:60], r[1], r[2], r[3], r[4], r[5], 100.0 * r[5] / tot))
    except Exception as e:  # noqa
        out.append("no kernel table: %r" % e)
    try:
        cc = cols(c, "counters_collection")
        kn = [k for k in cc if "kernel" in k and "name" in k][0] if any("kernel" in k and "name" in k for k in cc) else "name"
        rows = c.execute("select %s, counter_name, count(*), avg(value), sum(value) from counters_collection "
                         "group by %s, counter_name order by %s" % (kn, kn, kn)).fetchall()
        if rows:
            out.append("PMC (per dispatch avg): kernel | counter | dispatches | avg | sum")
            for r in rows:
                if "gcn" in r[0] or "gat" in r[0] or "combine" in r[0]:
                    out.append("  %-50s | %-28s | %5d | %16.1f | %18.1f" % (r[0][:50], r[1], r[2], r[3], r[4]))
    except Exception as e:  # noqa
        out.append("no counters: %r" % e)
    return "\n".join(out)


if __name__ == "__main__":
    for d in sys.argv[1:]:
        for db in sorted(glob.glob(os.path.join(d, "**", "*.db"), recursive=True)):
            print("==", db)
            print(summarize(db))
