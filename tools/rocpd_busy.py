"""GPU busy / idle time inside the marker window of a profile_step trace (union of kernel intervals):
   python tools/rocpd_busy.py trace.db <steps>"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
marks = [r[0] for r in db.execute("select start from kernels where name like '%spin_kernel%' order by start").fetchall()]
lo, hi = marks[-2], marks[-1]
cols = [r[1] for r in db.execute("pragma table_info(kernels)").fetchall()]
qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
rows = db.execute(f"select start, end{', ' + qcol if qcol else ''} from kernels where start > {lo} and start < {hi} "
                  f"and name not like '%spin_kernel%' order by start").fetchall()
def union(iv):
    busy, cur_s, cur_e = 0, None, None
    for s, e in iv:
        if cur_e is None or s > cur_e:
            if cur_e is not None: busy += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    if cur_e is not None: busy += cur_e - cur_s
    return busy
span = hi - lo
b = union([(r[0], r[1]) for r in rows])
print(f"window {span/1e6/steps:.2f} ms/step; some kernel running {b/1e6/steps:.2f} ms/step; GPU idle {(span-b)/1e6/steps:.2f} ms/step; "
      f"kernel time summed {sum(r[1]-r[0] for r in rows)/1e6/steps:.2f} ms/step")
if qcol:
    qs = {}
    for r in rows: qs.setdefault(r[2], []).append((r[0], r[1]))
    for q, iv in sorted(qs.items(), key=lambda kv: -len(kv[1])):
        print(f"  {qcol} {q}: {len(iv)/steps:.0f} launches/step, busy {union(iv)/1e6/steps:.2f} ms/step")
gaps = sorted(((rows[i+1][0] - max(r[1] for r in rows[max(0,i-8):i+1])) for i in range(len(rows)-1)), reverse=True)
big = [g for g in gaps if g > 20000]
print(f"gaps > 20 us: {len(big)/steps:.0f}/step totalling {sum(big)/1e6/steps:.2f} ms/step; largest {[round(g/1e3) for g in gaps[:8]]} us")
