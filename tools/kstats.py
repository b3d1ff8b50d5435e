#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd database (kernel-trace) into a per-kernel stats table (like --stats)."""
import sqlite3, sys, re
db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(rocpd_kernel_dispatch)")]
q = """select s.kernel_name, count(*), sum(d.end-d.start), avg(d.end-d.start), min(d.end-d.start), max(d.end-d.start)
       from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id
       group by s.kernel_name order by 3 desc"""
rows = list(cur.execute(q))
tot = sum(r[2] for r in rows)
print("%-90s %7s %12s %10s %10s %10s %6s" % ("kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "pct"))
for name, n, t, a, mn, mx in rows:
    name = re.sub(r"\s+", " ", name)[:90]
    print("%-90s %7d %12.1f %10.2f %10.2f %10.2f %6.2f" % (name, n, t / 1e3, a / 1e3, mn / 1e3, mx / 1e3, 100.0 * t / tot))
