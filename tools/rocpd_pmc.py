"""Per-kernel average of one PMC counter from a rocprofv3 rocpd database (`rocprofv3 --pmc X --kernel-trace`).
Usage: python tools/rocpd_pmc.py <db> [counter-name-substring]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
views = [r[0] for r in cur.execute("select name from sqlite_master where type='view'")]
cols = [r[1] for r in cur.execute("pragma table_info(counters_collection)")]
if not cols:
    print("no counters_collection view; views:", views)
    sys.exit(1)
name_col = "kernel_name" if "kernel_name" in cols else [c for c in cols if "kernel" in c and "name" in c][0]
cnt_col = "counter_name" if "counter_name" in cols else [c for c in cols if "counter" in c and "name" in c][0]
val_col = "value" if "value" in cols else [c for c in cols if "value" in c][0]
disp_col = "dispatch_id" if "dispatch_id" in cols else None
if disp_col:
    q = (f"select {name_col}, {cnt_col}, count(*), avg(v) from (select {name_col}, {cnt_col}, {disp_col}, sum({val_col}) as v "
         f"from counters_collection group by {name_col}, {cnt_col}, {disp_col}) group by {name_col}, {cnt_col} order by avg(v) desc")
else:
    q = f"select {name_col}, {cnt_col}, count(*), avg({val_col}) from counters_collection group by {name_col}, {cnt_col}"
print(f"{'kernel':64s} {'counter':14s} {'dispatches':>10s} {'avg per dispatch':>18s}")
for name, cnt, n, avg in cur.execute(q):
    if len(sys.argv) > 2 and sys.argv[2] not in cnt:
        continue
    print(f"{name[:64]:64s} {cnt:14s} {n:10d} {avg:18.1f}")
