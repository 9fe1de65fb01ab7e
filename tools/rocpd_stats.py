"""Per-kernel summary (calls, total, average, min, max in us) from a rocprofv3 rocpd database
(`rocprofv3 --kernel-trace --stats` writes <name>_results.db on ROCm 7.2).
Usage: python tools/rocpd_stats.py gpurun_out/prof/x_results.db [> profiles/NAME.txt]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
rows = cur.execute(f"select {name_col}, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                   f"from kernels group by {name_col} order by sum(end-start) desc").fetchall()
total = sum(r[2] for r in rows) or 1
print(f"{'kernel':70s} {'calls':>6s} {'total_us':>12s} {'avg_us':>10s} {'min_us':>10s} {'max_us':>10s} {'%':>6s}")
for name, calls, tot, avg, mn, mx in rows:
    short = name if len(name) <= 70 else name[:67] + "..."
    print(f"{short:70s} {calls:6d} {tot/1e3:12.1f} {avg/1e3:10.2f} {mn/1e3:10.2f} {mx/1e3:10.2f} {100*tot/total:6.2f}")
