"""fp32 head tail (grouped 3x3 64->3, 42 groups) at B=4, 180x180: per-kernel times vs the library formulations."""
import os as _os; _os.environ.setdefault("UD_RANDOM_INIT", "1")   # synthetic weights (tools never train for real)
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "cvpr2023-unidistill_amd")]
import torch
from unidistill_amd import _lib
from unidistill_amd.ops import head_tail_f32 as h
d = torch.device("cuda:0")
B, H, W, G, KM = 4, 180, 180, 42, 3
a = torch.randn(B, G * 64, H, W, device=d).contiguous(memory_format=torch.channels_last).requires_grad_(True)
w = (torch.randn(G * KM, 64, 3, 3, device=d) * 0.05).requires_grad_(True)
b = torch.zeros(G * KM, device=d, requires_grad=True)
z = h.group_tail(a, w, b, G, KM); gz = torch.randn_like(z)
for _ in range(2):
    z = h.group_tail(a, w, b, G, KM); z.backward(gz)
torch.cuda.synchronize()
_lib.prof_enable(True)
for _ in range(5):
    z = h.group_tail(a, w, b, G, KM); z.backward(gz)
torch.cuda.synchronize(); _lib.prof_enable(False)
gb = a.numel() * 4 / 1e9
for k in ("k_gtail_fwd", "k_gtail_dgrad", "k_gtail_wgrad"):
    ms, n = _lib.prof_read("head_tail." + k)
    t = ms / max(n, 1)
    print(f"{k:16s} {t*1e3:8.1f} us   ({gb/t:6.2f} TB/s over the 1.39 GB hidden tensor)")

# the BatchNorm-fused block: relu(bn(y)) -> tail, backward down to y
gamma = torch.rand(G * 64, device=d).add_(0.5).requires_grad_(True)
beta = (torch.randn(G * 64, device=d) * 0.1).requires_grad_(True)
rm, rv = torch.zeros(G * 64, device=d), torch.ones(G * 64, device=d)
for fused in (True, False):
  h.FUSED_BN_BWD = fused; print('fused BN backward' if fused else 'separate dgrad + BN backward')
  for i in range(7):
      if i == 2:
          torch.cuda.synchronize(); _lib.prof_enable(True)
      z = h.bn_relu_group_tail(a, gamma, beta, rm, rv, True, 0.1, 1e-5, None, None, w, b, G, KM); z.backward(gz)
  torch.cuda.synchronize(); _lib.prof_enable(False)
  for k in ("head_tail.k_gtail_fwd", "head_tail.k_gtail_wgrad", "head_tail.k_gtail_bn_bwd_reduce",
            "head_tail.k_gtail_bn_bwd_dx", "head_tail.k_gtail_dgrad", "bn_act.k_bwd_reduce", "bn_act.k_bwd_dx"):
      ms, n = _lib.prof_read(k)
      if n:
          t = ms / n
          print(f"{k:34s} {t*1e3:8.1f} us   ({gb/t:6.2f} TB/s over 1.39 GB)")
