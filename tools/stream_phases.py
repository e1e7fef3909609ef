"""Where the streams of a distillation step are on the GPU clock, WITHOUT a tracer (rocprofv3 adds ~8 us of host time per launch,
which turns the GPU-bound step into a host-bound one and changes what overlaps): HIP events recorded on the stream that runs each
phase -- student forward (main stream), teacher pass (its own stream), losses + backward (main; weight gradients on a third) --
and read back after the loop.
    B=4 CL=1 python tools/stream_phases.py            (AC=bf16 for the mixed-precision step)"""
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
marks = []
REC = [False]


def mark(name):
    if REC[0]:
        e = torch.cuda.Event(enable_timing=True); e.record()         # on the CURRENT stream
        marks.append((name, time.perf_counter(), e))


orig_teacher = step.teacher
def teacher(batch, prep, lidar_prepared=None):
    mark("teacher start")
    out = orig_teacher(batch, prep, lidar_prepared)
    mark("teacher end")
    return out
step.teacher = teacher
orig_model_fwd = step.model.forward
def model_fwd(*a, **k):
    mark("student fwd start")
    out = orig_model_fwd(*a, **k)
    mark("student fwd end")
    return out
step.model.forward = model_fwd
for i in range(4):
    tr.step(batch)
torch.cuda.synchronize()
REC[0] = True
e0 = torch.cuda.Event(enable_timing=True); e0.record(); torch.cuda.synchronize(); h0 = time.perf_counter()
for i in range(steps):
    tr.opt.zero_grad(set_to_none=True)
    mark("step start")
    if ac is not None:
        with torch.autocast("cuda", dtype=ac):
            out = tr.module(batch)
    else:
        out = tr.module(batch)
    mark("losses end")
    out["loss"].backward()
    mark("backward end")
    torch.nn.utils.clip_grad_norm_(tr.params, tr.grad_clip, foreach=True)
    tr.opt.step()
    mark("optimizer end")
torch.cuda.synchronize()
n = len(marks) // steps
last = marks[-2 * n:]
t_ref = e0.elapsed_time(last[0][2])
print(f"{'phase':>18} {'host ms':>9} {'gpu ms':>9}   (last two steps; ms since the first of them started on the GPU)")
for name, h, e in last:
    print(f"{name:>18} {(h - h0) * 1e3 - t_ref:9.2f} {e0.elapsed_time(e) - t_ref:9.2f}")
