"""Quick GPU timing of bev_pool fwd/bwd at the BASELINE shape with a synthetic 6-camera rig."""
import os as _os; _os.environ.setdefault("UD_RANDOM_INIT", "1")   # synthetic weights (tools never train for real)
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "cvpr2023-unidistill_amd")]
import numpy as np, torch
from unidistill_amd import synthetic as syn
from unidistill_amd.ops import bev_pool as bp

d = torch.device("cuda:0")
B = int(os.environ.get("B", 1)); ncam = int(os.environ.get("NCAM", 6)); C = 256
g = syn.rng()
s2e, intr, ida, bda = syn.camera_rig(g, B, ncam)
geom, _ = syn.frustum_bins_torch(s2e, intr, ida, bda, d)
N = geom.shape[1]
nx = ny = 180
feat = torch.randn(B, N, C, device=d)
out = torch.empty(B, ny, nx, C, device=d)
pos = torch.empty(B, N, 3, dtype=torch.int32, device=d)
def fwd(): bp._pool_fwd(geom, feat, out, pos, B, N, C, nx, ny, 1, bp.POOL_OVERWRITE)
fwd(); torch.cuda.synchronize()
kept = (pos[..., 0] >= 0)
cnt = torch.zeros(B * ny * nx, dtype=torch.long, device=d)
cell = (pos[..., 0].long() * ny + pos[..., 1].long()) * nx + pos[..., 2].long()
cnt.index_add_(0, cell[kept], torch.ones_like(cell[kept]))
print(f"N={N} kept={kept.float().mean().item():.3f} cells_nonempty={(cnt>0).sum().item()} max/cell={cnt.max().item()} "
      f"heavy(>64) cells={(cnt>64).sum().item()} pts_in_heavy={cnt[cnt>64].sum().item()}")
def timeit(fn, n=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
t = timeit(fwd)
alg = B * N * (12 + C * 4 + 12) + B * ny * nx * C * 4
print(f"bev_pool fwd: {t:.1f} us  algorithmic {alg/1e6:.1f} MB -> {alg/t/1e6:.2f} TB/s ({alg/t/1e6/8*100:.1f}% of 8 TB/s)")
gout = torch.randn(B, ny, nx, C, device=d).permute(0, 3, 1, 2)
def bwd(): bp._pool_bwd(gout, pos, B, N, C, nx, ny)
t = timeit(bwd)
algb = B * ny * nx * C * 4 + B * N * 12 + B * N * C * 4
print(f"bev_pool bwd (NHWC grad): {t:.1f} us  algorithmic {algb/1e6:.1f} MB -> {algb/t/1e6:.2f} TB/s")
gout2 = torch.randn(B, C, ny, nx, device=d)
def bwd2(): bp._pool_bwd(gout2, pos, B, N, C, nx, ny)
t = timeit(bwd2)
print(f"bev_pool bwd (NCHW grad, staged): {t:.1f} us -> {algb/t/1e6:.2f} TB/s")
