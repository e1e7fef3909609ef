"""Which ATen ops does one fp32 distillation step launch, and from where?  torch.profiler over one step, grouped by op and the
innermost repo frame; prints launch counts (the step has ~900 small library launches of 2-6 us each)."""
import collections, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "cvpr2023-unidistill_amd")]
os.environ.setdefault("UD_RANDOM_INIT", "1")
import torch
from torch.profiler import profile, ProfilerActivity
from unidistill_amd import train as T
dev = torch.device("cuda:0")
torch.manual_seed(0)
tr = T.Trainer(T.DistillStep("camera_exp_distill_lidar"), device=dev, autocast_dtype=None, channels_last=True)
batch = T.synthetic_batch(dev, 4)
for _ in range(3):
    tr.step(batch)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    tr.step(batch)
    torch.cuda.synchronize()
rows = []
for ka in prof.key_averages(group_by_stack_n=12):
    if not ka.key.startswith("aten::") or ka.device_time_total <= 0:
        continue
    where = "?"
    for fr in ka.stack:
        if "unidistill_amd/" in fr and "torch/" not in fr:
            where = fr.split("unidistill_amd/")[-1][:80]
            break
    rows.append((ka.count, ka.self_device_time_total / 1e3, ka.key, where))
agg = collections.defaultdict(lambda: [0, 0.0])
for n, t, k, w in rows:
    agg[(k, w)][0] += n
    agg[(k, w)][1] += t
print("aten ops with device time, by (op, innermost package frame):")
for (k, w), (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:70]:
    print(f"{n:5d}  {t:7.3f} ms  {k:28s} {w}")
