"""Full names + launch counts of ATen elementwise kernels inside the marker window of a profile_step trace."""
import re, sqlite3, sys
db = sqlite3.connect(sys.argv[1]); steps = int(sys.argv[2])
marks = [r[0] for r in db.execute("select start from kernels where name like '%spin_kernel%' order by start").fetchall()]
where = f" where start > {marks[-2]} and start < {marks[-1]}" if len(marks) >= 2 else ""
rows = db.execute(f"select name, count(*), sum(duration), avg(grid_x) from kernels{where} group by name").fetchall() if False else \
       db.execute(f"select name, count(*), sum(duration) from kernels{where} group by name").fetchall()
for name, n, dur in sorted(rows, key=lambda r: -r[2]):
    if "elementwise" in name or "Functor" in name:
        short = re.sub(r"at::native::|\(anonymous namespace\)::|c10::", "", name)
        print(f"{dur/1e6/steps:7.3f} ms/step {n/steps:6.1f} x  {short[:260]}")
