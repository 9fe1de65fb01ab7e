"""One step's kernel timeline from a rocprofv3 rocpd database: every launch between two consecutive launches of an
anchor kernel, with its start offset, duration and the idle gap before it.
Usage: python tools/rocpd_timeline.py <db> <anchor-substring> [which-occurrence]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
rows = cur.execute(f"select {name_col}, start, end from kernels order by start").fetchall()
anchor = sys.argv[2]
which = int(sys.argv[3]) if len(sys.argv) > 3 else 5
idx = [k for k, r in enumerate(rows) if anchor in r[0]]
if len(idx) < which + 2:
    print("anchor occurs", len(idx), "times")
    sys.exit(1)
a, b = idx[which], idx[which + 1]
t0 = rows[a][1]
prev_end = rows[a][1]
print(f"{'kernel':72s} {'start_us':>9s} {'dur_us':>8s} {'gap_us':>8s}")
busy = 0
for name, st, en in rows[a:b]:
    short = name if len(name) <= 72 else name[:69] + "..."
    print(f"{short:72s} {(st - t0) / 1e3:9.2f} {(en - st) / 1e3:8.2f} {(st - prev_end) / 1e3:8.2f}")
    busy += en - st
    prev_end = en
print(f"step (anchor to anchor): {(rows[b][1] - t0) / 1e3:.2f} us, kernels busy {busy / 1e3:.2f} us, {b - a} launches")
