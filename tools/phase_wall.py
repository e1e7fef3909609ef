"""Wall-clock time per phase of the distillation step (device-synchronised around each phase): which parts
are fixed (launch-bound) and which scale with the batch.  B=1 vs B=4 side by side."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "cvpr2023-unidistill_amd")]
import torch
from unidistill_amd import train
dev = torch.device("cuda:0")
res = {}
for B in (1, 4):
    torch.manual_seed(0)
    step = train.DistillStep("camera_exp_distill_lidar").to(dev)
    step.overlap_teacher = False
    batch = train.synthetic_batch(dev, B, sweeps=1)
    tr = train.Trainer(step, device=dev, autocast_dtype=torch.bfloat16, channels_last=True)
    for _ in range(3): tr.step(batch)
    m = tr.module
    acc = {}
    def run(name, fn):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        out = fn()
        torch.cuda.synchronize(); acc[name] = acc.get(name, 0.0) + (time.perf_counter() - t0) * 1e3
        return out
    N = 5
    for _ in range(N):
        tr.opt.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            prep = run("prep (targets, masks)", lambda: m.prep(batch))
            prep = run("reduce normalisers", lambda: m.reduce(prep))
            tout = run("teacher forward", lambda: m.teacher(batch, prep))
            out = run("student forward + losses", lambda: m.student_loss(batch, prep, tout))
        run("backward", lambda: out["loss"].backward())
        run("clip + optimizer", lambda: (torch.nn.utils.clip_grad_norm_(tr.params, tr.grad_clip, foreach=True), tr.opt.step()))
    res[B] = {k: v / N for k, v in acc.items()}
    del tr, step
for k in res[1]:
    print(f"{k:28s} B=1 {res[1][k]:7.2f} ms   B=4 {res[4][k]:7.2f} ms   per-sample slope {(res[4][k]-res[1][k])/3:6.2f}   fixed {res[1][k]-(res[4][k]-res[1][k])/3:6.2f}")
print(f"{'total':28s} B=1 {sum(res[1].values()):7.2f} ms   B=4 {sum(res[4].values()):7.2f} ms")
