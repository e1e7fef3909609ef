"""A few launches of ud_conv3x3_nhwc_bf16 on the trunk shape 256 -> 128 @180 x 180 x 4 (the target of the FETCH_SIZE / WRITE_SIZE
passes behind roofline_mfma.traffic): algorithmic bytes = x 66.4 MB + y 33.2 MB + w 0.6 MB = 100.1 MB per launch."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "cvpr2023-unidistill_amd")]
import torch
from unidistill_amd.ops import conv2d as c
d = torch.device("cuda:0")
B, ci, co, H, W = 4, 256, 128, 180, 180
x = torch.randn(B, ci, H, W, device=d).bfloat16().contiguous(memory_format=torch.channels_last)
w = (torch.randn(co, ci, 3, 3, device=d) * 0.03)
wt = c.tap_major(w)
scrub = torch.empty(512 << 20, dtype=torch.uint8, device=d)
for _ in range(6):
    scrub.zero_()                      # x out of the Infinity Cache
    c._launch(x, wt, co)
torch.cuda.synchronize()
print("done")
