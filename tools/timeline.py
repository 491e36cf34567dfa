"""Print the kernel timeline of the last steps of a rocprofv3 kernel trace (rocpd sqlite).
usage: python tools/timeline.py gpurun_out/prof/xxx_results.db [n_kernels]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
n_show = int(sys.argv[2]) if len(sys.argv) > 2 else 24
rows = db.execute("select name,start,end,stream_id from kernels order by start").fetchall()
t0 = rows[0][1]
prev_end = {}
for nm, s, e, st in rows[len(rows) - n_show - 8:len(rows) - 8]:
    print("%-36s stream %3s start %10.1f end %10.1f dur %7.1f" % (nm[:36], st, (s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3))
