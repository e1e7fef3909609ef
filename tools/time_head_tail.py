"""Head tail (BN -> ReLU -> 42 per-head convs) at the nuScenes BEV size: HIP kernels vs the library path."""
import os as _os; _os.environ.setdefault("UD_RANDOM_INIT", "1")   # synthetic weights (tools never train for real)
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "cvpr2023-unidistill_amd")]
import torch
from unidistill_amd import _lib
from unidistill_amd.ops import head_tail as ht
dev = torch.device("cuda:0")
B = int(os.environ.get("B", 4)); G, kmax, H, W = 42, 3, 180, 180
y = torch.randn(B, G * 64, H, W, device=dev, dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last).requires_grad_(True)
gamma = torch.ones(G * 64, device=dev, requires_grad=True); beta = torch.zeros(G * 64, device=dev, requires_grad=True)
w2 = (torch.randn(G * kmax, 64, 3, 3, device=dev) * 0.05).requires_grad_(True); b2 = torch.zeros(G * kmax, device=dev, requires_grad=True)
rm, rv = torch.zeros(G * 64, device=dev), torch.ones(G * 64, device=dev)
gz = torch.randn(B, G * kmax, H, W, device=dev)
def step():
    y.grad = None
    z = ht.head_tail(y, gamma, beta, w2, b2, rm, rv, True, 0.1, 1e-5, G, kmax)
    z.backward(gz)
for _ in range(3): step()
torch.cuda.synchronize()
_lib.prof_enable(True)
for _ in range(10): step()
torch.cuda.synchronize()
_lib.prof_enable(False)
nbytes = y.numel() * 2
print(f"B={B}: hidden tensor {nbytes/1e6:.0f} MB")
for k in ("head_tail.stats", "head_tail.k_fwd", "head_tail.k_wgrad", "head_tail.k_bn_sums", "head_tail.k_dy"):
    ms, n = _lib.prof_read(k)
    us = ms / max(n, 1) * 1e3
    passes = 2 if k.endswith("k_dy") else 1
    print(f"  {k:22s} {us:8.1f} us   {passes*nbytes/us/1e6:6.2f} TB/s of y traffic")
