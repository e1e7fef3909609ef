"""Winograd F(4x4,3x3) fp32 weight-gradient kernel (csrc/conv2d_f32_wino4_wgrad.hip): error against an fp64 weight gradient and
time against the F(2x2) kernel at the 3x3 shapes of the distillation step."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "cvpr2023-unidistill_amd")]
os.environ.setdefault('UD_RANDOM_INIT', '1')
import torch
from unidistill_amd.ops import conv2d_f32 as c
d = torch.device("cuda:0")
def t(fn, n=10):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
shapes = [("small 32->64 @13x19 x2", 2, 32, 64, 13, 19), ("small 64->64 @16x16", 1, 64, 64, 16, 16),
          ("trunk 256->128 @180^2", 4, 256, 128, 180, 180), ("trunk 128->128 @180^2", 4, 128, 128, 180, 180),
          ("trunk 256->256 @90^2", 4, 256, 256, 90, 90), ("head shared 512->64 @180^2", 4, 512, 64, 180, 180),
          ("head first 64->2688 @180^2", 4, 64, 2688, 180, 180), ("resnet 64->64 @64x176 x24", 24, 64, 64, 64, 176),
          ("resnet 128->128 @32x88 x24", 24, 128, 128, 32, 88), ("resnet 256->256 @16x44 x24", 24, 256, 256, 16, 44),
          ("resnet 512->512 @8x22 x24", 24, 512, 512, 8, 22)]
only = sys.argv[1:]
for name, B, ci, co, H, W in shapes:
    if only and not any(o in name for o in only): continue
    torch.manual_seed(0)
    x = torch.randn(B, ci, H, W, device=d).contiguous(memory_format=torch.channels_last)
    gy = torch.randn(B, co, H, W, device=d).contiguous(memory_format=torch.channels_last)
    w = torch.randn(co, ci, 3, 3, device=d)
    fl = 2 * B * H * W * co * 9 * ci
    ref = torch.nn.grad.conv2d_weight(x.double(), w.shape, gy.double(), padding=1) if fl < 1e12 else None
    res = {}
    for tag, f4 in (("F2", False), ("F4", True)):
        c.USE_WINO4_WGRAD, c.WINO4_WGRAD_ALL = f4, True
        gw = c.weight_grad(x, gy, w, 3)
        torch.cuda.synchronize()
        err = float((gw.double() - ref).abs().max() / ref.abs().max()) if ref is not None else float("nan")
        res[tag] = (t(lambda: c.weight_grad(x, gy, w, 3)), err)
    print(f"wgrad {name:30s} " + "  ".join(f"{k} {v[0]:8.1f} us ({fl/v[0]/1e6:6.1f} TF) err {v[1]:.1e}" for k, v in res.items()), flush=True)
