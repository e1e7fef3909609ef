"""Representative plain 1x1 fp32 launches (forward line kernel + weight gradient) for the SQ counter passes:
   bash tools/pmc_kernel.sh k_conv1x1 python tools/pmc_f32_1x1.py [B Cin Cout H W]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "cvpr2023-unidistill_amd")]
os.environ.setdefault('UD_RANDOM_INIT', '1')
import torch
from unidistill_amd.ops import conv2d_f32 as c
B, ci, co, H, W = [int(v) for v in sys.argv[1:6]] if len(sys.argv) > 5 else (24, 1024, 256, 16, 44)
d = torch.device("cuda:0")
x = torch.randn(B, ci, H, W, device=d).contiguous(memory_format=torch.channels_last)
gy = torch.randn(B, co, H, W, device=d).contiguous(memory_format=torch.channels_last)
w = torch.randn(co, ci, 1, 1, device=d) * 0.03
w2 = w.reshape(co, ci).contiguous()
for _ in range(13):
    c._launch1(x, w2, co)
    c.weight_grad(x, gy, w, 1)
torch.cuda.synchronize()
