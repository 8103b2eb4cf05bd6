"""Dump the per-kernel summary of a rocprofv3 (rocpd sqlite) capture as CSV:
    python tools/rocpd_summary.py gpurun_out/prof/x_results.db > profiles/x_kernel_stats.csv
Columns: kernel, calls, total_us, avg_us, min_us, max_us, pct."""
import csv
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = db.execute(
    "select name, count(*), sum(duration)/1e3, avg(duration)/1e3, min(duration)/1e3, max(duration)/1e3 "
    "from kernels group by name order by sum(duration) desc").fetchall()
total = sum(r[2] for r in rows) or 1.0
w = csv.writer(sys.stdout)
w.writerow(["kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "pct"])
for name, calls, tot, avg, mn, mx in rows:
    short = name.replace("(anonymous namespace)::", "").replace("void ", "")
    if len(short) > 110:
        short = short[:107] + "..."
    w.writerow([short, calls, f"{tot:.1f}", f"{avg:.1f}", f"{mn:.1f}", f"{mx:.1f}", f"{100*tot/total:.2f}"])
