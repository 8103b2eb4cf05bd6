"""Dump the per-kernel summary of a rocprofv3 (rocpd sqlite) capture as CSV:
    python tools/rocpd_summary.py gpurun_out/prof/x_results.db > profiles/x_kernel_stats.csv
Columns: kernel, grid, calls, total_us, avg_us, min_us, max_us, pct.  Launches of one kernel are kept apart by grid
size when the capture records it (the coarse and the fine network launch the same kernels over 262,144 and 786,432
rows), so that an average can be compared with the per-launch HIP-event time bench.py reports."""
import csv
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute("pragma table_info(kernels)").fetchall()]
grid = next((c for c in ("grid_size", "grid_size_x", "grid_x", "grid") if c in cols), None)
gexpr = grid if grid else "0"
rows = db.execute(
    f"select name, {gexpr}, count(*), sum(duration)/1e3, avg(duration)/1e3, min(duration)/1e3, max(duration)/1e3 "
    f"from kernels group by name, {gexpr} order by sum(duration) desc").fetchall()
total = sum(r[3] for r in rows) or 1.0
w = csv.writer(sys.stdout)
w.writerow(["kernel", "grid", "calls", "total_us", "avg_us", "min_us", "max_us", "pct"])
for name, g, calls, tot, avg, mn, mx in rows:
    short = name.replace("(anonymous namespace)::", "").replace("void ", "")
    if len(short) > 110:
        short = short[:107] + "..."
    w.writerow([short, g if grid else "", calls, f"{tot:.1f}", f"{avg:.1f}", f"{mn:.1f}", f"{mx:.1f}", f"{100*tot/total:.2f}"])
if not grid:
    print("# (no grid-size column in this capture: columns = " + ", ".join(cols) + ")", file=sys.stderr)
