"""Is the sparse 128 -> 128 kernel bound by its gathers?  The same kernel on synthetic rulebooks with the real pair density:
  real-like random: neighbours are random rows;  local: neighbours are rows o + small offsets (a tile's rows come from a window)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "cvpr2023-unidistill_amd")]
import torch
from unidistill_amd.ops import spconv as sp
M, K, C = 198615, 27, 128
dev = torch.device("cuda")
torch.manual_seed(0)
feat = torch.randn(M, C, device=dev)
w = torch.randn(C, K, C, device=dev) * 0.05          # [n][k][c]
ws = (K * C, C, 1)
def timeit(nbr):
    for _ in range(2): sp._conv(feat, nbr, w, ws, False, None, C, C)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): sp._conv(feat, nbr, w, ws, False, None, C, C)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 5 * 1e3
base = torch.arange(M, device=dev, dtype=torch.int32)[:, None]
for name, dens in (("all 27 offsets", 1.0), ("65 % of the offsets (the real layers' density)", 0.65)):
    keep = (torch.rand(M, K, device=dev) < dens)
    keep[:, 13] = True
    for kind in ("identity (every offset gathers the row itself: one 512 B row per output row)",
                 "window (row o + k - 13: 27 neighbours inside a 27-row window)",
                 "random rows of the whole tensor"):
        if kind.startswith("identity"): nb = base.expand(M, K).clone()
        elif kind.startswith("window"): nb = (base + torch.arange(K, device=dev, dtype=torch.int32)[None] - 13).clamp(0, M - 1)
        else: nb = torch.randint(0, M, (M, K), device=dev, dtype=torch.int32)
        nbr = torch.where(keep, nb, torch.full_like(nb, -1)).contiguous()
        pairs = int(keep.sum())
        us = timeit(nbr)
        print(f"{name:48s} {kind[:60]:62s} pairs {pairs/1e6:5.2f} M  {us:8.1f} us  {2.0*pairs*C*C/us/1e6:6.1f} TFLOP/s  gathered {pairs*512/us/1e6:5.2f} TB/s")
