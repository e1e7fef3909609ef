"""Print per-kernel PMC counter averages from a rocprofv3 rocpd .db (name filter optional)."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); flt = sys.argv[2] if len(sys.argv) > 2 else ""
tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
view = "counters_collection" if "counters_collection" in tabs else None
cols = [r[1] for r in db.execute(f"pragma table_info('{view}')")]
print("columns:", cols)
kcol = "kernel_name" if "kernel_name" in cols else ("name" if "name" in cols else cols[0])
ccol = "counter_name" if "counter_name" in cols else "name"
vcol = "value" if "value" in cols else "counter_value"
rows = db.execute(f"select {kcol}, {ccol}, count(*), avg({vcol}), min({vcol}), max({vcol}) from {view} group by {kcol}, {ccol}").fetchall()
for r in rows:
    if flt in str(r[0]):
        print(f"{str(r[0])[:70]:70s} {r[1]:14s} n={r[2]} avg={r[3]:.1f} min={r[4]:.1f} max={r[5]:.1f}")
