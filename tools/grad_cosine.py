"""Root-causing tool for run-to-run gradient agreement of the distillation step.

Runs the same step (same seed, same batch) several times and prints, in forward order,
  * the first module whose OUTPUT differs bitwise between two identical bf16 runs (forward determinism),
  * per-parameter gradient cosine: bf16 vs bf16 (same seed), fp32 vs fp32, bf16 vs fp32.
    WL=camera_exp_distill_lidar B=1 python tools/grad_cosine.py
"""
import os as _os; _os.environ.setdefault("UD_RANDOM_INIT", "1")   # synthetic weights (tools never train for real)
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "cvpr2023-unidistill_amd")]
import torch
from unidistill_amd import train
dev = torch.device("cuda:0")
WL = os.environ.get("WL", "camera_exp_distill_lidar")
B = int(os.environ.get("B", 1))
DET = os.environ.get("DET", "0") == "1"
if DET:
    torch.backends.cudnn.deterministic = True
    torch.use_deterministic_algorithms(True, warn_only=True)


def run(ac, overlap=False, record=True, hip=True):
    from unidistill_amd.layers import dense
    dense.Conv2d.hip_enabled = hip          # False: every dense conv / BN through the libraries (torch autocast)
    torch.manual_seed(0)
    step = train.DistillStep(WL).to(dev).train()
    if os.environ.get("TAME"):
        # well-conditioned variant: damp every residual branch (last BN gamma of a block) so that rounding
        # noise is not amplified layer by layer as it is in a randomly initialised 50-layer network
        with torch.no_grad():
            for n, m in step.named_modules():
                if n.endswith(".bn3") or n.endswith(".bn2") and "backbone_3d" in n:
                    m.weight.mul_(float(os.environ["TAME"]))
    step.overlap_teacher = overlap
    train.to_channels_last(step)
    batch = train.synthetic_batch(dev, B)
    acts = []
    hooks = []
    if record:
        for n, m in step.model.named_modules():
            if not list(m.children()):
                def hk(mod, inp, out, n=n):
                    t = out.features if hasattr(out, "features") else out
                    if torch.is_tensor(t):
                        acts.append((n, t.detach().float().flatten()[:: max(1, t.numel() // 65536)].clone(),
                                     float(t.detach().float().abs().mean())))
                hooks.append(m.register_forward_hook(hk))
    if ac is not None:
        with torch.autocast("cuda", dtype=ac):
            out = step(batch)
    else:
        out = step(batch)
    out["loss"].backward()
    torch.cuda.synchronize()
    for h in hooks:
        h.remove()
    grads = [(n, p.grad.flatten().float().clone()) for n, p in step.model.named_parameters() if p.grad is not None]
    return float(out["loss"]), grads, acts, {k: float(v) for k, v in out["tb"].items() if torch.is_tensor(v) and v.numel() == 1}


def cos(a, b):
    return float(torch.nn.functional.cosine_similarity(a.double(), b.double(), dim=0))


la, ga, aa, tba = run(torch.bfloat16)
lb, gb, ab, tbb = run(torch.bfloat16)
print("bf16 loss", la, lb)
first = None
for (n, x, m), (_, y, _) in zip(aa, ab):
    if not torch.equal(x, y):
        first = n
        rel = float((x - y).abs().max() / (x.abs().max() + 1e-30))
        print(f"FIRST forward difference between two identical bf16 runs: {n}  max rel diff {rel:.3e}")
        break
if first is None:
    print("forward activations of two identical bf16 runs: bitwise identical (sampled)")
lc, gc, ac_, tbc = run(None)
ld, gd, ad, tbd = run(None)
print("fp32 loss", lc, ld)
for (n, x, m), (_, y, _) in zip(ac_, ad):
    if not torch.equal(x, y):
        print(f"FIRST forward difference between two identical fp32 runs: {n}")
        break
else:
    print("forward activations of two identical fp32 runs: bitwise identical (sampled)")
for k in tba:
    print(f"  {k:20s} bf16 {tba[k]:12.5f} {tbb[k]:12.5f}   fp32 {tbc.get(k, float('nan')):12.5f}")
le, ge, ae, tbe = run(torch.bfloat16, hip=False)
print("bf16 LIBRARY path (torch autocast, no hand-written dense kernels) loss", le)
print("activation relative deviation vs fp32 by module (every 8th common module): ours-bf16 / library-bf16")
fa, fl = {n: x for n, x, _ in aa}, {n: x for n, x, _ in ae}
k = 0
for n, y, _ in ac_:
    if n in fa and n in fl and fa[n].shape == y.shape == fl[n].shape:
        if k % 8 == 0:
            print(f"   {float((fa[n] - y).norm() / (y.norm() + 1e-30)):9.3e} {float((fl[n] - y).norm() / (y.norm() + 1e-30)):9.3e}  {n}")
        k += 1
gmax = max(float(g.norm()) for _, g in gc)
print(f"{'param':70s} {'|g| fp32':>10s} {'cos bf-bf':>9s} {'cos 32-32':>9s} {'cos bf-32':>9s} {'cos lib-32':>10s}")
tot = {k: [] for k in ("bb", "ff", "bf", "lf")}
for i, ((n, a), (_, b), (_, c), (_, d), (_, e)) in enumerate(zip(ga, gb, gc, gd, ge)):
    gn = float(c.norm())
    cbb, cff, cbf, clf = cos(a, b), cos(c, d), cos(a, c), cos(e, c)
    big = gn > 1e-3 * gmax
    if big:
        tot["bb"].append(cbb); tot["ff"].append(cff); tot["bf"].append(cbf); tot["lf"].append(clf)
    if os.environ.get("ALL") or (big and min(cbb, cff, cbf) < 0.99 and i % 6 == 0):
        print(f"{n:70s} {gn:10.3e} {cbb:9.4f} {cff:9.4f} {cbf:9.4f} {clf:10.4f}{'' if big else '  (tiny)'}")
allg = lambda g: torch.cat([x for _, x in g])
print("whole-gradient cos: bf-bf %.5f  32-32 %.5f  bf-32 %.5f  lib-32 %.5f" % (cos(allg(ga), allg(gb)), cos(allg(gc), allg(gd)), cos(allg(ga), allg(gc)), cos(allg(ge), allg(gc))))
for k, v in tot.items():
    v = sorted(v)
    print(k, "large-norm params: n=%d min %.4f  p10 %.4f median %.4f" % (len(v), v[0], v[len(v) // 10], v[len(v) // 2]))
