"""Which source lines of the package issue the ATen ops of one fp32 distillation step?  A TorchDispatchMode counts every op that
reaches the dispatcher by its innermost package frame (forward; ops issued by the autograd engine are grouped under the
backward node that runs them).  Launch counts, not device time: the step has ~700 small library launches of 2-6 us each."""
import collections, os, sys, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "cvpr2023-unidistill_amd")]
os.environ.setdefault("UD_RANDOM_INIT", "1")
import torch
from torch.utils._python_dispatch import TorchDispatchMode
from unidistill_amd import train as T
dev = torch.device("cuda:0")
torch.manual_seed(0)
AC = torch.bfloat16 if os.environ.get("AC") == "bf16" else None       # AC=bf16: the mixed-precision step
tr = T.Trainer(T.DistillStep("camera_exp_distill_lidar"), device=dev, autocast_dtype=AC, channels_last=True)
batch = T.synthetic_batch(dev, 4)
for _ in range(3):
    tr.step(batch)
torch.cuda.synchronize()
SKIP = ("aten.view", "aten._unsafe_view", "aten.detach", "aten.alias", "aten.t.", "aten.transpose", "aten.permute", "aten.slice",
        "aten.select", "aten.expand", "aten.unsqueeze", "aten.squeeze", "aten.as_strided", "aten.reshape", "aten._reshape_alias",
        "aten.empty", "aten.unbind", "aten.split", "aten.narrow", "aten.stride", "aten.size", "aten.is_", "aten.lift_fresh",
        "aten.sym_", "prim.", "aten.new_empty", "aten.empty_like", "aten.empty_strided", "aten._local_scalar_dense", "aten.item")
cnt = collections.Counter()
special = collections.Counter()
engine = collections.Counter()
FILTER = [f for f in os.environ.get("OPS", "").split(",") if f]      # OPS=index_put,sort: list these ops with their sites


class Count(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if not name.startswith(SKIP):
            site = "(autograd engine / optimizer)"
            for fr in reversed(traceback.extract_stack()[:-1]):
                if "unidistill_amd/" in fr.filename and "tools/" not in fr.filename:
                    site = f"{fr.filename.split('unidistill_amd/')[-1]}:{fr.lineno} {fr.name}"
                    break
            cnt[site] += 1
            if site.startswith("(autograd"):
                engine[name] += 1
            if FILTER and any(f in name for f in FILTER):
                shp = tuple(args[0].shape) if args and isinstance(args[0], torch.Tensor) else (tuple(args[0]) if args and isinstance(args[0], (list, tuple)) else ())
                special[(name + " " + str(shp), site)] += 1
        return func(*args, **(kwargs or {}))


with Count():
    tr.step(batch)
torch.cuda.synchronize()
tot = sum(cnt.values())
print(f"{tot} dispatched (non-view) ops in one step; by source line:")
by_fn = collections.Counter()
for s, n in cnt.items():
    by_fn[s.split(" ")[0].split(":")[0] + " " + s.split(" ")[-1]] += n
for s, n in by_fn.most_common(40):
    print(f"{n:5d}  {s}")
for (name, site), n in special.most_common(40):
    print(f"{n:5d}  {name:60s} {site}")
print("ops issued outside a package frame (autograd engine, optimizer):")
for name, n in engine.most_common(25):
    print(f"{n:5d}  {name}")
