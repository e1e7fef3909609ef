"""One F(4x4) forward shape for the SQ counter passes (tools/pmc_kernel.sh k_conv3x3_wino4_f32 python tools/pmc_wino4.py [B Cin Cout H W])."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "cvpr2023-unidistill_amd")]
os.environ.setdefault('UD_RANDOM_INIT', '1')
import torch
from unidistill_amd.ops import conv2d_f32 as c
B, ci, co, H, W = [int(v) for v in sys.argv[1:6]] if len(sys.argv) > 5 else (4, 128, 128, 180, 180)
d = torch.device("cuda:0")
x = torch.randn(B, ci, H, W, device=d).contiguous(memory_format=torch.channels_last)
w = (torch.randn(co, ci, 3, 3, device=d) * 0.03).contiguous(memory_format=torch.channels_last)
c.WINO4_MIN_FILL = 0.0
for _ in range(13):
    c._launch3(x, w)
torch.cuda.synchronize()
