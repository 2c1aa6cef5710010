#!/usr/bin/env python3
"""Turns a rocprofv3 rocpd SQLite database (--kernel-trace --stats, default output format) into the per-kernel summary
text committed under profiles/.  Usage: python profiles/summarize_rocpd.py <results.db> [> profiles/<name>.txt]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
tables = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
disp = [t for t in tables if t.startswith("rocpd_kernel_dispatch")][0]
sym = [t for t in tables if t.startswith("rocpd_info_kernel_symbol")][0]
rows = list(cur.execute(
    f"select k.kernel_name, count(*), sum(d.end-d.start), avg(d.end-d.start), min(d.end-d.start), max(d.end-d.start), "
    f"max(d.grid_size_x), max(d.workgroup_size_x), max(d.group_segment_size) "
    f"from {disp} d join {sym} k on d.kernel_id=k.id group by k.kernel_name order by 3 desc"))
total = sum(r[2] for r in rows) or 1
print("%-100s %6s %12s %12s %12s %12s %6s %10s %6s %8s" % ("kernel", "calls", "total_ms", "avg_ms", "min_ms", "max_ms",
                                                          "pct", "grid", "wg", "lds_B"))
for name, n, tot, avg, mn, mx, grid, wg, lds in rows:
    print("%-100s %6d %12.4f %12.4f %12.4f %12.4f %6.2f %10d %6d %8d" % (name[:100], n, tot / 1e6, avg / 1e6, mn / 1e6,
                                                                      mx / 1e6, 100.0 * tot / total, grid, wg, lds or 0))
