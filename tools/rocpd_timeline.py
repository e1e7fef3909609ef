"""Per-stream timeline of ONE steady-state training step from a tools/profile_step.py trace (rocprofv3 --kernel-trace):
   python tools/rocpd_timeline.py trace.db [bin_ms]

Replaces "HBM-bound work hides beside MFMA-bound kernels" with numbers: for the last complete step of the marker window
 * per stream (HIP queue): launches, busy ms (union of its kernel intervals), share of the step;
 * pairwise overlap of the streams' busy intervals, the union (some kernel running) and GPU idle;
 * the same split by what bounds the kernel: "matrix" (convolution / sparse-convolution / weight-gradient kernels on the MFMA pipe)
   versus "stream" (BatchNorm, head tail, elementwise, reductions, optimizer ...): time with NO matrix kernel in flight is the part
   of the step the MFMA pipe sits out;
 * a bin-by-bin table (default 2 ms): busy fraction per stream and the kernel that held most of the bin on each.
Step boundaries: the fused AdamW launch (multi_tensor_apply ... FusedAdam) that ends every Trainer.step."""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
bin_ms = float(sys.argv[2]) if len(sys.argv) > 2 else 2.0
marks = [r[0] for r in db.execute("select start from kernels where name like '%spin_kernel%' order by start").fetchall()]
lo, hi = marks[-2], marks[-1]
cols = [r[1] for r in db.execute("pragma table_info(kernels)").fetchall()]
qcol = "stream_id" if "stream_id" in cols else ("queue_id" if "queue_id" in cols else None)
if qcol is None:
    raise SystemExit("no stream / queue column in the kernels view")
rows = db.execute(f"select start, end, {qcol}, name from kernels where start > {lo} and start < {hi} "
                  f"and name not like '%spin_kernel%' order by start").fetchall()
adam = [r[1] for r in rows if "FusedAdam" in r[3] or "fused_adam" in r[3].lower()]
# several multi-tensor launches per optimizer step: a step ends at the last of a burst
ends = [t for i, t in enumerate(adam) if i + 1 == len(adam) or adam[i + 1] - t > 5e6]
if len(ends) < 2:
    raise SystemExit("fewer than two optimizer steps in the window")
s0, s1 = ends[-2], ends[-1]
step = [r for r in rows if s0 < r[0] <= s1 or (r[0] <= s0 < r[1])]
span = s1 - s0


def matrix(name):
    n = name
    return any(k in n for k in ("k_conv", "k_wino", "k_wgrad", "wgrad_f32", "k_stem", "k_spconv", "mfma", "k_gtail_wgrad",
                                "Cijk", "gemm"))


def union(iv):
    iv = sorted(iv)
    out, cs, ce = [], None, None
    for s, e in iv:
        if ce is None or s > ce:
            if ce is not None:
                out.append((cs, ce))
            cs, ce = s, e
        else:
            ce = max(ce, e)
    if ce is not None:
        out.append((cs, ce))
    return out


def total(iv):
    return sum(e - s for s, e in iv)


def inter(a, b):
    i = j = 0
    out = []
    while i < len(a) and j < len(b):
        s, e = max(a[i][0], b[j][0]), min(a[i][1], b[j][1])
        if s < e:
            out.append((s, e))
        if a[i][1] < b[j][1]:
            i += 1
        else:
            j += 1
    return out


def clip(iv):
    return [(max(s, s0), min(e, s1)) for s, e in iv if min(e, s1) > max(s, s0)]


streams = {}
for s, e, q, n in step:
    streams.setdefault(q, []).append((s, e, n))
order = sorted(streams, key=lambda q: -total(union(clip([(s, e) for s, e, _ in streams[q]]))))
names = {q: "S%d" % i for i, q in enumerate(order)}
busy = {q: union(clip([(s, e) for s, e, _ in streams[q]])) for q in order}
allu = union([iv for q in order for iv in busy[q]])
ms = 1e-6
print(f"step {span * ms:.2f} ms (between the last two optimizer launches of the window); {len(step)} launches")
print()
print("| stream | launches | busy ms | % of step | matrix-kernel ms | streaming-kernel ms | top kernels (ms) |")
print("|---|---|---|---|---|---|---|")
for q in order:
    ks = {}
    for s, e, n in streams[q]:
        short = n.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:48]
        ks[short] = ks.get(short, 0) + (min(e, s1) - max(s, s0))
    top = ", ".join(f"{k} {v * ms:.1f}" for k, v in sorted(ks.items(), key=lambda kv: -kv[1])[:4])
    mt = total(union(clip([(s, e) for s, e, n in streams[q] if matrix(n)])))
    st = total(union(clip([(s, e) for s, e, n in streams[q] if not matrix(n)])))
    print(f"| {names[q]} ({qcol} {q}) | {len(streams[q])} | {total(busy[q]) * ms:.2f} | {100 * total(busy[q]) / span:.0f} | "
          f"{mt * ms:.2f} | {st * ms:.2f} | {top} |")
print()
print(f"some kernel running: {total(allu) * ms:.2f} ms; GPU idle: {(span - total(allu)) * ms:.2f} ms; "
      f"kernel time summed over streams: {sum(total(busy[q]) for q in order) * ms:.2f} ms")
for i, a in enumerate(order):
    for b in order[i + 1:]:
        ov = total(inter(busy[a], busy[b]))
        if ov * ms >= 0.05:
            print(f"overlap {names[a]} & {names[b]}: {ov * ms:.2f} ms")
mat = union(clip([(s, e) for s, e, q, n in step if matrix(n)]))
stm = union(clip([(s, e) for s, e, q, n in step if not matrix(n)]))
both = total(inter(mat, stm))
print(f"a matrix kernel in flight: {total(mat) * ms:.2f} ms; a streaming kernel in flight: {total(stm) * ms:.2f} ms; both at once: "
      f"{both * ms:.2f} ms; ONLY streaming kernels (MFMA pipe sits out): {(total(stm) - both) * ms:.2f} ms; "
      f"only matrix kernels: {(total(mat) - both) * ms:.2f} ms")
# which streaming kernels run with no matrix kernel beside them
alone = {}
for s, e, q, n in step:
    if matrix(n):
        continue
    iv = clip([(s, e)])
    if not iv:
        continue
    a = total(iv) - total(inter(iv, mat))
    short = n.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:60]
    alone[short] = alone.get(short, 0) + a
print()
print("streaming kernels by the time they run with NO matrix kernel beside them (ms per step):")
for k, v in sorted(alone.items(), key=lambda kv: -kv[1])[:14]:
    print(f"  {v * ms:6.2f}  {k}")
print()
nb = int(span * ms / bin_ms) + 1
print(f"| t (ms) | " + " | ".join(f"{names[q]} busy %, main kernel" for q in order[:4]) + " |")
print("|---|" + "---|" * min(4, len(order)))
for b in range(nb):
    b0, b1 = s0 + b * bin_ms / ms, min(s0 + (b + 1) * bin_ms / ms, s1)
    if b1 <= b0:
        break
    cells = []
    for q in order[:4]:
        ks, tot = {}, 0
        for s, e, n in streams[q]:
            o = min(e, b1) - max(s, b0)
            if o > 0:
                short = n.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0].split("<")[0][:28]
                ks[short] = ks.get(short, 0) + o
        u = total(inter(busy[q], [(b0, b1)]))
        cells.append(f"{100 * u / (b1 - b0):3.0f} {max(ks, key=ks.get) if ks else ''}")
    print(f"| {b * bin_ms:5.1f} | " + " | ".join(cells) + " |")
