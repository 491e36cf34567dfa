"""Summarise a rocprofv3 (rocpd sqlite) kernel trace: per-kernel calls / total / avg / min / max.
usage: python tools/rocpd_stats.py gpurun_out/prof/xxx_results.db [> profiles/xxx_kernel_stats.txt]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
rows = cur.execute("select %s, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels "
                   "group by %s order by sum(end-start) desc" % (name_col, name_col)).fetchall()
tot = sum(r[2] for r in rows) or 1
print("%-58s %8s %12s %10s %10s %10s %6s" % ("kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "pct"))
for n, c, s, a, mn, mx in rows:
    print("%-58s %8d %12.1f %10.2f %10.2f %10.2f %6.2f" % (n[:58], c, s / 1e3, a / 1e3, mn / 1e3, mx / 1e3, 100.0 * s / tot))
