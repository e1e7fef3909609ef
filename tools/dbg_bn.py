import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "cvpr2023-unidistill_amd")]
import torch, torch.nn.functional as F
from unidistill_amd.layers.dense import batchnorm_act
from unidistill_amd import _lib
torch.manual_seed(0)
for (B, C, H, W, res) in [(24, 64, 64, 176, False), (4, 128, 180, 180, False), (24, 256, 64, 176, True), (24, 2048, 8, 22, True)]:
    x = (torch.randn(B, C, H, W, device="cuda") * 2 + 0.5).bfloat16().contiguous(memory_format=torch.channels_last)
    r = torch.randn(B, C, H, W, device="cuda").bfloat16().contiguous(memory_format=torch.channels_last) if res else None
    gy = torch.randn(B, C, H, W, device="cuda").bfloat16().contiguous(memory_format=torch.channels_last)
    bn = torch.nn.BatchNorm2d(C).cuda(); bn2 = torch.nn.BatchNorm2d(C).cuda()
    xa = x.clone().requires_grad_(True); xb = x.float().requires_grad_(True)
    ra = r.clone().requires_grad_(True) if res else None; rb = r.float().requires_grad_(True) if res else None
    y = batchnorm_act(bn, xa, ra, True); y.backward(gy)
    yr = bn2(xb); yr = F.relu(yr + rb if res else yr); yr.backward(gy.float())
    def e(a, b): return ((a.float() - b).norm() / b.norm()).item()
    print(B, C, H, W, res, "y", e(y, yr), "dx", e(xa.grad, xb.grad), "dg", e(bn.weight.grad, bn2.weight.grad), "db", e(bn.bias.grad, bn2.bias.grad),
          "dres", e(ra.grad, rb.grad) if res else None, "rm", e(bn.running_mean, bn2.running_mean), "rv", e(bn.running_var, bn2.running_var))
    _lib.prof_enable(True)
    for _ in range(5):
        xa.grad = None
        y = batchnorm_act(bn, xa, ra, True); y.backward(gy)
    torch.cuda.synchronize(); _lib.prof_enable(False)
    nb = x.numel() * 2
    for k in ("bn_act.stats", "bn_act.k_fwd", "bn_act.k_bwd_reduce", "bn_act.k_bwd_dx"):
        ms, n = _lib.prof_read(k); us = max(ms / max(n, 1) * 1e3, 1e-3)
        print(f"   {k:22s} {us:8.1f} us  ({nb/us/1e6:.2f} TB/s per tensor pass)")
