"""Step time with DistributedDataParallel forced around the module on ONE rank (RCCL communicator of one rank, bucket views,
communication hook of ops/wgrad_stream.py) against the plain single-process step: what the N > 1 launch mode costs before the
first byte crosses xGMI.      python tools/time_ddp_step.py        (AC=bf16 for the mixed-precision step)"""
import os as _os; _os.environ.setdefault("UD_RANDOM_INIT", "1")
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "cvpr2023-unidistill_amd")]
import torch
import torch.distributed as dist
DDP = os.environ.get("DDP", "1") == "1"
if DDP:
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=os.environ.get("MASTER_PORT", "29571"), RANK="0", WORLD_SIZE="1",
                      UD_FORCE_DDP="1")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
from unidistill_amd import train
from unidistill_amd.ops import wgrad_stream
dev = torch.device("cuda:0")
torch.manual_seed(0)
ac = {"bf16": torch.bfloat16, "": None}[os.environ.get("AC", "")]
tr = train.Trainer(train.DistillStep("camera_exp_distill_lidar"), device=dev, autocast_dtype=ac, channels_last=True)
assert (tr.ddp is not None) == DDP
batch = train.synthetic_batch(dev, 4)
for i in range(6):
    o = tr.step(batch)
    if os.environ.get("LOSSES"):
        print("warm", i, float(o["loss"].detach()).hex())
torch.cuda.synchronize()
n = int(os.environ.get("STEPS", 20))
t0 = time.perf_counter()
for _ in range(n):
    out = tr.step(batch)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
print(f"DDP={int(DDP)} wgrad_stream={wgrad_stream.state()} {wgrad_stream.STATS}: {dt * 1e3:.2f} ms/step  loss={float(out['loss'].detach()):.4f}")
if DDP:
    dist.destroy_process_group()
