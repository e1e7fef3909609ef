"""Which kernels follow a given helper kernel (e.g. MIOpen's SubTensorOp set/cast) in a rocpd trace:
   python tools/rocpd_neighbors.py trace.db SubTensorOp"""
import collections, sqlite3, sys
db = sqlite3.connect(sys.argv[1]); pat = sys.argv[2]
marks = [r[0] for r in db.execute("select start from kernels where name like '%spin_kernel%' order by start").fetchall()]
where = f" where start > {marks[-2]} and start < {marks[-1]}" if len(marks) >= 2 else ""
rows = db.execute(f"select name, duration, grid_x from kernels{where} order by start").fetchall() if False else \
       db.execute(f"select name, duration from kernels{where} order by start").fetchall()
after, before = collections.Counter(), collections.Counter()
for i, (name, dur) in enumerate(rows):
    if pat in name:
        j = i + 1
        while j < len(rows) and pat in rows[j][0]: j += 1
        k = i - 1
        while k >= 0 and pat in rows[k][0]: k -= 1
        after[(name[:28], rows[j][0][:90] if j < len(rows) else "-")] += 1
        before[(name[:28], rows[k][0][:90] if k >= 0 else "-")] += 1
print("== followed by")
for (a, b), n in after.most_common(25): print(f"{n:5d}  {a:28s} -> {b}")
print("== preceded by")
for (a, b), n in before.most_common(25): print(f"{n:5d}  {b} -> {a}")
