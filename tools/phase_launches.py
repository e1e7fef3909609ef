"""Kernel launches and host/GPU time per phase of the distillation step (torch.profiler)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "cvpr2023-unidistill_amd")]
import torch
from torch.profiler import profile, ProfilerActivity
from unidistill_amd import train
dev = torch.device("cuda:0"); B = int(os.environ.get("B", 4))
torch.manual_seed(0)
step = train.DistillStep("camera_exp_distill_lidar").to(dev)
batch = train.synthetic_batch(dev, B, sweeps=1)
tr = train.Trainer(step, device=dev, autocast_dtype=torch.bfloat16, channels_last=True)
for _ in range(3): tr.step(batch)
torch.cuda.synchronize()
def run(name, fn):
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        out = fn()
        torch.cuda.synchronize()
    ev = prof.key_averages()
    launches = sum(e.count for e in ev if e.key in ("hipLaunchKernel", "hipExtModuleLaunchKernel", "hipModuleLaunchKernel"))
    gpu = sum(getattr(e, "self_device_time_total", 0) for e in ev) / 1e3
    cpu = sum(e.self_cpu_time_total for e in ev) / 1e3
    print(f"{name:28s} launches {launches:5d}   gpu {gpu:7.2f} ms   cpu {cpu:7.2f} ms")
    return out
m = tr.module
tr.opt.zero_grad(set_to_none=True)
with torch.autocast("cuda", dtype=torch.bfloat16):
    prep = run("prep (targets, masks)", lambda: m.prep(batch))
    prep = run("reduce normalisers", lambda: m.reduce(prep))
    tout = run("teacher forward", lambda: m.teacher(batch, prep))
    out = run("student forward + losses", lambda: m.student_loss(batch, prep, tout))
run("backward", lambda: out["loss"].backward())
run("clip + optimizer", lambda: (torch.nn.utils.clip_grad_norm_(tr.params, tr.grad_clip, foreach=True), tr.opt.step()))
