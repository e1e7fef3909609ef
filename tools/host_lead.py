"""How far the host runs ahead of the GPU inside a training step: host timestamps at the phase boundaries of Trainer.step
(forward enqueued / backward enqueued / optimizer enqueued) against HIP events recorded at the same points.
    B=4 CL=1 python tools/host_lead.py"""
import os as _os; _os.environ.setdefault("UD_RANDOM_INIT", "1")
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "cvpr2023-unidistill_amd")]
import torch
from unidistill_amd import train
B = int(os.environ.get("B", 4)); steps = int(os.environ.get("STEPS", 8))
dev = torch.device("cuda:0")
torch.manual_seed(0)
step = train.DistillStep("camera_exp_distill_lidar")
batch = train.synthetic_batch(dev, B)
ac = {"bf16": torch.bfloat16, "": None}[os.environ.get("AC", "")]
tr = train.Trainer(step, device=dev, autocast_dtype=ac, channels_last=os.environ.get("CL", "1") == "1")
for i in range(3):
    tr.step(batch)
torch.cuda.synchronize()
marks = []


def mark(name):
    e = torch.cuda.Event(enable_timing=True); e.record()
    marks.append((name, time.perf_counter(), e))


e0 = torch.cuda.Event(enable_timing=True); e0.record(); torch.cuda.synchronize(); h0 = time.perf_counter()
for i in range(steps):
    tr.opt.zero_grad(set_to_none=True)
    mark("start")
    if ac is not None:
        with torch.autocast("cuda", dtype=ac):
            out = tr.module(batch)
    else:
        out = tr.module(batch)
    mark("fwd")
    out["loss"].backward()
    mark("bwd")
    torch.nn.utils.clip_grad_norm_(tr.params, tr.grad_clip, foreach=True)
    mark("clip")
    tr.opt.step()
    mark("opt")
torch.cuda.synchronize()
print(f"{'phase':>6} {'host ms':>9} {'gpu ms':>9} {'lead ms':>9}   (times since the loop start; lead = gpu - host)")
for name, h, e in marks[-10:]:
    g = e0.elapsed_time(e)
    print(f"{name:>6} {(h - h0) * 1e3:9.2f} {g:9.2f} {g - (h - h0) * 1e3:9.2f}")
