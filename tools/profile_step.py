"""Run a few training steps of a workload (for rocprofv3 / timing breakdowns)."""
import os as _os; _os.environ.setdefault("UD_RANDOM_INIT", "1")   # synthetic weights (tools never train for real)
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "cvpr2023-unidistill_amd")]
import torch
from unidistill_amd import train
wl = os.environ.get("WL", "camera_exp_distill_lidar")
B = int(os.environ.get("B", 1)); steps = int(os.environ.get("STEPS", 5))
dev = torch.device("cuda:0")
torch.manual_seed(0)
if wl in ("lidar", "camera", "fusion"):
    step = train.DetectStep(wl)
    batch = train.synthetic_batch(dev, B, with_imgs=wl != "lidar", with_points=wl != "camera", sweeps=int(os.environ.get("SWEEPS", 1)))
else:
    step = train.DistillStep(wl)
    batch = train.synthetic_batch(dev, B, sweeps=int(os.environ.get("SWEEPS", 1)))
ac = {"bf16": torch.bfloat16, "": None}[os.environ.get("AC", "")]
tr = train.Trainer(step, device=dev, autocast_dtype=ac, channels_last=os.environ.get("CL", "0") == "1")
for i in range(3):
    out = tr.step(batch)
    if os.environ.get('LOSSES'): print('warm', i, float(out['loss']))
torch.cuda._sleep(1000); torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(steps):
    out = tr.step(batch)
    if os.environ.get('LOSSES'): print('step', i, float(out['loss']))
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
torch.cuda._sleep(1000); torch.cuda.synchronize()
print(f"{wl} B={B}: {dt*1e3:.1f} ms/step -> {B/dt:.2f} samples/s  loss={out['loss'].item():.3f}  mem={torch.cuda.max_memory_allocated()/2**30:.1f} GiB")
