"""Time one shape of the persistent conv kernel (UD_CONV_P_ABL=n selects a timing ablation: results are wrong)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "cvpr2023-unidistill_amd")]
import torch
from unidistill_amd.ops import conv2d as c2
from unidistill_amd import _lib
_lib.load().ud_conv3x3_persistent(int(os.environ.get('UD_CONV_P', '1')))
dev = torch.device("cuda:0")
B, ci, H, W, co = (int(v) for v in os.environ.get("SHAPE", "4,128,180,180,128").split(","))
x = torch.randn(B, ci, H, W, device=dev, dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
wt = c2.tap_major(torch.randn(co, ci, 3, 3, device=dev) * 0.02)
for _ in range(3): c2._launch(x, wt, co)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): c2._launch(x, wt, co)
e1.record(); torch.cuda.synchronize()
t = e0.elapsed_time(e1) / 20 * 1e3
print(f"ABL={os.environ.get('UD_CONV_P_ABL', '0'):3s} {t:8.1f} us  {2 * B * H * W * co * ci * 9 / t / 1e6:6.0f} TFLOP/s-equivalent")
