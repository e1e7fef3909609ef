"""Count ATen ops / kernel launches per training step by Python source region (torch.profiler, CPU side)."""
import os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "cvpr2023-unidistill_amd")]
import torch
from torch.profiler import profile, ProfilerActivity, record_function
from unidistill_amd import train
dev = torch.device("cuda:0"); B = int(os.environ.get("B", 4))
torch.manual_seed(0)
step = train.DistillStep("camera_exp_distill_lidar")
batch = train.synthetic_batch(dev, B, sweeps=1)
tr = train.Trainer(step, device=dev, autocast_dtype=torch.bfloat16, channels_last=True)
for _ in range(3): tr.step(batch)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=False, record_shapes=False) as prof:
    tr.step(batch)
    torch.cuda.synchronize()
ev = prof.key_averages()
rows = sorted(ev, key=lambda e: -e.count)
print("top ops by call count (one step):")
for e in rows[:45]:
    print(f"  {e.count:5d}  cpu {e.self_cpu_time_total/1e3:7.2f} ms  cuda {getattr(e, 'self_device_time_total', 0)/1e3:7.2f} ms  {e.key[:90]}")
print("total cpu self time (ms):", sum(e.self_cpu_time_total for e in ev) / 1e3)
