"""A few launches of ud_conv3x3_nhwc_bf16 for the SQ counter passes (tools/pmc_kernel.sh k_conv3x3 python tools/pmc_conv_p.py):
SHAPE=B,Cin,H,W,Cout (default: trunk 128 -> 128 @180 x 180 x 4)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "cvpr2023-unidistill_amd")]
import torch
from unidistill_amd.ops import conv2d as c
from unidistill_amd import _lib
_lib.load().ud_conv3x3_persistent(int(os.environ.get('UD_CONV_P', '1')))
d = torch.device("cuda:0")
B, ci, H, W, co = (int(v) for v in os.environ.get("SHAPE", "4,128,180,180,128").split(","))
x = torch.randn(B, ci, H, W, device=d).bfloat16().contiguous(memory_format=torch.channels_last)
w = (torch.randn(co, ci, 3, 3, device=d) * 0.03)
wt = c.tap_major(w)
for _ in range(6):
    c._launch(x, wt, co)
torch.cuda.synchronize()
print("done")
