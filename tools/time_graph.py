import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "cvpr2023-unidistill_amd")]
import torch
from unidistill_amd import train
dev = torch.device("cuda:0")
ac = {"bf16": torch.bfloat16, "": None}[os.environ.get("AC", "")]
for B in [int(b) for b in os.environ.get("BS", "1,2,4").split(",")]:
    batch = train.synthetic_batch(dev, batch_size=B, ncam=6)
    torch.manual_seed(0)
    g = train.GraphTrainer(train.DistillStep(os.environ.get("WL", "camera_exp_distill_lidar")), batch, device=dev, autocast_dtype=ac)
    for _ in range(3): g.step(batch)
    torch.cuda._sleep(1000); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): o = g.step(batch)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10; torch.cuda._sleep(1000); torch.cuda.synchronize()
    print(f"graph B={B} ac={ac}: {dt*1e3:.1f} ms/step -> {B/dt:.2f} samples/s loss={o['loss'].item():.2f} mem={torch.cuda.max_memory_allocated()/2**30:.1f}GiB", flush=True)
    del g
