"""Winograd F(4x4,3x3) fp32 kernel (csrc/conv2d_f32_wino4.hip): error against an fp64 convolution and time against F(2x2,3x3) /
the direct kernel at the 3x3 shapes of the distillation step (forward and data-gradient orientation)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "cvpr2023-unidistill_amd")]
os.environ.setdefault('UD_RANDOM_INIT', '1')
import torch, torch.nn.functional as F
from unidistill_amd.ops import conv2d_f32 as c
d = torch.device("cuda:0")
def t(fn, n=10):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
shapes3 = [("tiny 8->64 @16x32", 1, 8, 64, 16, 32), ("odd 24->40 @13x19 x2", 2, 24, 40, 13, 19),
           ("trunk 256->128 @180^2", 4, 256, 128, 180, 180), ("trunk 128->128 @180^2", 4, 128, 128, 180, 180),
           ("trunk 256->256 @90^2", 4, 256, 256, 90, 90), ("head shared 512->64 @180^2", 4, 512, 64, 180, 180),
           ("head first 64->2688 @180^2", 4, 64, 2688, 180, 180), ("head dgrad 2688->64 @180^2", 4, 2688, 64, 180, 180),
           ("resnet 64->64 @64x176 x24", 24, 64, 64, 64, 176),
           ("resnet 128->128 @32x88 x24", 24, 128, 128, 32, 88), ("resnet 256->256 @16x44 x24", 24, 256, 256, 16, 44),
           ("resnet 512->512 @8x22 x24", 24, 512, 512, 8, 22)]
only = sys.argv[1:]
for name, B, ci, co, H, W in shapes3:
    if only and not any(o in name for o in only): continue
    torch.manual_seed(0)
    x = torch.randn(B, ci, H, W, device=d).contiguous(memory_format=torch.channels_last)
    w = torch.randn(co, ci, 3, 3, device=d) * (2.0 / (9 * ci)) ** 0.5
    wl = w.contiguous(memory_format=torch.channels_last)
    fl = 2 * B * H * W * co * 9 * ci
    ref = F.conv2d(x.double(), w.double(), None, 1, 1) if fl < 3e12 else None
    res = {}
    for tag, f2, f4 in (("direct", False, False), ("F2", True, False), ("F4", True, True)):
        c.USE_WINOGRAD, c.USE_WINO4 = f2, f4
        c.WINO4_MIN_FILL = 0.0
        try:
            y = c._launch3(x, wl)
        except RuntimeError:
            res[tag] = (float("nan"), float("nan"))
            continue
        torch.cuda.synchronize()
        err = float((y.double() - ref).abs().max() / ref.abs().max()) if ref is not None else float("nan")
        res[tag] = (t(lambda: c._launch3(x, wl)), err)
    # data-gradient orientation of the same parameter (transposed, taps reversed)
    c.USE_WINOGRAD, c.USE_WINO4 = True, True
    gy = torch.randn(B, co, H, W, device=d).contiguous(memory_format=torch.channels_last)
    gx = c._launch3(gy, wl, transposed=True)
    gref = F.conv_transpose2d(gy.double(), w.double(), None, 1, 1) if fl < 3e12 else None
    gerr = float((gx.double() - gref).abs().max() / gref.abs().max()) if gref is not None else float("nan")
    print(f"3x3 {name:30s} " + "  ".join(f"{k} {v[0]:8.1f} us ({fl/v[0]/1e6:6.1f} TF) err {v[1]:.1e}" for k, v in res.items())
          + f"  | F4 dgrad err {gerr:.1e}", flush=True)
