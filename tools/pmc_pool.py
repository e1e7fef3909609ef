"""Run ud_bev_pool_fwd a few times (cache scrubbed) for rocprofv3 --pmc passes."""
import os as _os; _os.environ.setdefault("UD_RANDOM_INIT", "1")   # synthetic weights (tools never train for real)
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "cvpr2023-unidistill_amd")]
import torch
from unidistill_amd import synthetic as syn
from unidistill_amd.ops import bev_pool as bp
d = torch.device("cuda:0")
B, C, nx, ny = 1, 256, 180, 180
s2e, intr, ida, bda = syn.camera_rig(syn.rng(7), B, 6)
geom, _ = syn.frustum_bins_torch(s2e, intr, ida, bda, d)
N = geom.shape[1]
feat = torch.randn(B, N, C, device=d); out = torch.empty(B, ny, nx, C, device=d)
pos = torch.empty(B, N, 3, dtype=torch.int32, device=d)
scrub = torch.empty(512 << 20, dtype=torch.uint8, device=d)
for _ in range(5):
    scrub.zero_()
    bp._pool_fwd(geom, feat, out, pos, B, N, C, nx, ny, 1, bp.POOL_OVERWRITE)
torch.cuda.synchronize()
