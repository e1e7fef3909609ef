"""Sparse encoder fwd+bwd under bf16 autocast: kernel time of conv (fwd+dgrad) and wgrad launches."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "cvpr2023-unidistill_amd")]
import torch
from unidistill_amd import synthetic as syn, config as C, _lib
from unidistill_amd.layers.lidar import LidarEncoder
dev = torch.device("cuda:0")
B = int(os.environ.get("B", 4)); sweeps = int(os.environ.get("SWEEPS", 1))
enc = LidarEncoder(C.LIDAR_ENCODER).to(dev).train()
g = syn.rng()
pts = [torch.from_numpy(syn.lidar_cloud(g, 30000, sweeps)).to(dev) for _ in range(B)]
n = min(p.shape[0] for p in pts); pts = [p[:n] for p in pts]
def step():
    for p in enc.parameters(): p.grad = None
    with torch.autocast("cuda", dtype=torch.bfloat16):
        out = enc(pts)
    out.float().square().mean().backward()
for _ in range(3): step()
torch.cuda.synchronize()
_lib.prof_enable(True)
for _ in range(5): step()
torch.cuda.synchronize(); _lib.prof_enable(False)
for k in ("spconv.k_conv", "spconv.k_wgrad"):
    ms, n = _lib.prof_read(k)
    print(f"{k:16s} {ms/5:7.2f} ms/step over {n//5} launches")
