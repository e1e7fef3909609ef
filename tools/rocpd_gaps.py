"""Largest GPU-idle gaps inside the marker window of a profile_step trace, with the kernels on either side:
   python tools/rocpd_gaps.py trace.db [n]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); n = int(sys.argv[2]) if len(sys.argv) > 2 else 10
marks = [r[0] for r in db.execute("select start from kernels where name like '%spin_kernel%' order by start").fetchall()]
lo, hi = marks[-2], marks[-1]
rows = db.execute(f"select start, end, name from kernels where start > {lo} and start < {hi} and name not like '%spin_kernel%' order by start").fetchall()
gaps, cur_end, last = [], None, None
for s, e, name in rows:
    if cur_end is not None and s > cur_end:
        gaps.append((s - cur_end, last, name, (s - lo) / 1e6))
    if cur_end is None or e > cur_end:
        cur_end, last = e, name
for g, a, b, t in sorted(gaps, reverse=True)[:n]:
    print(f"{g/1e3:8.1f} us idle at +{t:7.2f} ms   after {a[:60]:60s} before {b[:60]}")
