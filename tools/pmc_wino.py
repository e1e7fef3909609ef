"""A few launches of the fp32 Winograd kernels (forward: ud_conv3x3_wino_nhwc_f32; weight gradient: ud_conv3x3_wino_wgrad_nhwc_f32)
on the trunk shape 128 -> 128 @180 x 180 x 4, for rocprofv3 --pmc passes (SQ busy / MFMA busy / wait counters)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "cvpr2023-unidistill_amd")]
os.environ.setdefault("UD_RANDOM_INIT", "1")
import torch
from unidistill_amd.ops import conv2d_f32 as c
d = torch.device("cuda:0")
B, ci, co, H, W = 4, 128, 128, 180, 180
x = torch.randn(B, ci, H, W, device=d).contiguous(memory_format=torch.channels_last)
gy = torch.randn(B, co, H, W, device=d).contiguous(memory_format=torch.channels_last)
w = torch.randn(co, ci, 3, 3, device=d) * 0.03
for _ in range(4):
    c._launch3(x, w)
    c.weight_grad(x, gy, w, 3)
torch.cuda.synchronize()
print("done")
