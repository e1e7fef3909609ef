"""Host-side profile of the distillation step (cProfile over a few steps; AC=bf16 for the mixed-precision step): where the
Python time of the ~1 400 launches per step goes.  The bf16 step is partly host-bound (GPU idle 15 % under the tracer)."""
import cProfile, os, pstats, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "cvpr2023-unidistill_amd")]
os.environ.setdefault("UD_RANDOM_INIT", "1")
import torch
from unidistill_amd import train as T
dev = torch.device("cuda:0")
torch.manual_seed(0)
AC = torch.bfloat16 if os.environ.get("AC") == "bf16" else None
tr = T.Trainer(T.DistillStep("camera_exp_distill_lidar"), device=dev, autocast_dtype=AC, channels_last=True)
batch = T.synthetic_batch(dev, 4)
for _ in range(5):
    tr.step(batch)
torch.cuda.synchronize()
N = int(os.environ.get("STEPS", 10))
pr = cProfile.Profile()
pr.enable()
for _ in range(N):
    tr.step(batch)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime")
print(f"per step (/{N}):")
rows = sorted(st.stats.items(), key=lambda kv: -kv[1][2])[:int(os.environ.get("TOP", 45))]
for (fn, line, name), (cc, nc, tt, ct, _callers) in rows:
    short = fn.split("unidistill_amd/")[-1] if "unidistill_amd/" in fn else os.path.basename(fn)
    print(f"{tt / N * 1e3:8.3f} ms self {ct / N * 1e3:8.3f} ms cum {nc / N:8.1f} calls  {short}:{line} {name}")
