"""Which ATen elementwise ops does one step run, on which shapes, and what do they cost on the GPU?

torch.profiler with record_shapes over one steady-state step; prints aten::add / add_ / copy_ / mul / fill_ / zero_ ...
grouped by input shapes, sorted by device time.  AC=bf16 for the mixed-precision step.
"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "cvpr2023-unidistill_amd")]
import torch
from torch.profiler import profile, ProfilerActivity
from unidistill_amd import train

B = int(os.environ.get("B", 4))
dev = torch.device("cuda:0")
torch.manual_seed(0)
step = train.DistillStep(os.environ.get("WL", "camera_exp_distill_lidar"))
batch = train.synthetic_batch(dev, B)
ac = {"bf16": torch.bfloat16, "": None}[os.environ.get("AC", "")]
tr = train.Trainer(step, device=dev, autocast_dtype=ac, channels_last=True)
for _ in range(3):
    tr.step(batch)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    tr.step(batch)
    torch.cuda.synchronize()
want = os.environ.get("OPS", "aten::add,aten::add_,aten::copy_,aten::mul,aten::mul_,aten::fill_,aten::zero_,aten::sub,aten::div,aten::cat,aten::sum").split(",")
rows = [e for e in prof.key_averages(group_by_input_shape=True) if e.key in want]
rows.sort(key=lambda e: -e.self_device_time_total)
tot = 0.0
for e in rows[: int(os.environ.get("TOP", 60))]:
    tot += e.self_device_time_total
    print(f"{e.self_device_time_total / 1e3:8.3f} ms {e.count:5d} x  {e.key:14s} {str(e.input_shapes)[:150]}")
print(f"listed: {tot / 1e3:.2f} ms")
