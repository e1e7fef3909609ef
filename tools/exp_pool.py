"""k_pool access-pattern experiment: real frustum geometry vs consecutive-rows-per-cell geometry."""
import os as _os; _os.environ.setdefault("UD_RANDOM_INIT", "1")   # synthetic weights (tools never train for real)
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "cvpr2023-unidistill_amd")]
import torch
from unidistill_amd import synthetic as syn, _lib
from unidistill_amd.ops import bev_pool as bp
d = torch.device("cuda:0")
B, C = 1, 256
g = syn.rng()
s2e, intr, ida, bda = syn.camera_rig(g, B, 6)
geom_real, _ = syn.frustum_bins_torch(s2e, intr, ida, bda, d)
N = geom_real.shape[1]
feat = torch.randn(B, N, C, device=d)
scrub = torch.empty(512 << 20, dtype=torch.uint8, device=d)
def run(geom, nx, ny, tag):
    out = torch.empty(B, ny, nx, C, device=d); pos = torch.empty(B, N, 3, dtype=torch.int32, device=d)
    for _ in range(2): bp._pool_fwd(geom, feat, out, pos, B, N, C, nx, ny, 1, bp.POOL_OVERWRITE)
    torch.cuda.synchronize(); _lib.prof_enable(True)
    for _ in range(8):
        scrub.zero_(); bp._pool_fwd(geom, feat, out, pos, B, N, C, nx, ny, 1, bp.POOL_OVERWRITE)
    torch.cuda.synchronize(); _lib.prof_enable(False)
    ms, n = _lib.prof_read("bev_pool.k_pool")
    kept = (pos[..., 0] >= 0).sum().item()
    us = ms / n * 1e3
    actual = kept * C * 4 + nx * ny * C * 4 + kept * 8
    print(f"{tag}: k_pool {us:.1f} us, kept {kept} ({kept/N:.2f}), actual bytes {actual/1e6:.0f} MB -> {actual/us/1e6:.2f} TB/s actual")
run(geom_real, 180, 180, "real frustum")
n = torch.arange(N, device=d)
cell = n // 12
seq = torch.stack([cell % 200, cell // 200, torch.zeros_like(cell)], -1).int().unsqueeze(0).contiguous()
run(seq, 200, 200, "12 consecutive rows per cell")
cell = n // 64
seq = torch.stack([cell % 100, cell // 100, torch.zeros_like(cell)], -1).int().unsqueeze(0).contiguous()
run(seq, 100, 100, "64 consecutive rows per cell")
perm = torch.randperm(N, device=d) // 12
rnd = torch.stack([perm % 200, perm // 200, torch.zeros_like(perm)], -1).int().unsqueeze(0).contiguous()
run(rnd, 200, 200, "12 random rows per cell")
