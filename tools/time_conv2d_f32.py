"""fp32 convolutions: hand-written fp32 MFMA kernels (ops/conv2d_f32.py) vs the library (MIOpen through torch),
forward and data gradient, at the shapes of the distillation step (B=4)."""
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
shapes3 = [("trunk 256->128 @180^2", 4, 256, 128, 180, 180), ("trunk 128->128 @180^2", 4, 128, 128, 180, 180),
           ("trunk 256->256 @90^2", 4, 256, 256, 90, 90), ("head shared 512->64 @180^2", 4, 512, 64, 180, 180),
           ("head first 64->2688 @180^2", 4, 64, 2688, 180, 180), ("resnet 64->64 @64x176 x24", 24, 64, 64, 64, 176),
           ("resnet 128->128 @32x88 x24", 24, 128, 128, 32, 88), ("resnet 256->256 @16x44 x24", 24, 256, 256, 16, 44)]
for name, B, ci, co, H, W in shapes3:
    x = torch.randn(B, ci, H, W, device=d).contiguous(memory_format=torch.channels_last)
    w = torch.randn(co, ci, 3, 3, device=d) * 0.03
    wt = w.permute(0, 2, 3, 1).contiguous()
    wl = w.contiguous(memory_format=torch.channels_last)
    fl = 2 * B * H * W * co * 9 * ci
    c.USE_WINOGRAD = False
    a = t(lambda: c._launch3(x, wl))
    c.USE_WINOGRAD = True
    a2 = t(lambda: c._launch3(x, wl)) if c.wino_pays(H, W, ci, co) else float("nan")
    b = t(lambda: F.conv2d(x, wl, None, 1, 1))
    print(f"3x3 {name:32s} direct {a:8.1f} us ({fl/a/1e6:6.1f} TF)   winograd {a2:8.1f} us ({fl/a2/1e6:6.1f} TF eff.)   library {b:8.1f} us ({fl/b/1e6:6.1f} TF)")
shapes1 = [("64->256 @64x176 x24", 24, 64, 256, 64, 176), ("256->64 @64x176 x24", 24, 256, 64, 64, 176),
           ("512->128 @32x88 x24", 24, 512, 128, 32, 88), ("1024->256 @16x44 x24", 24, 1024, 256, 16, 44),
           ("depth 512->368 @16x44 x24", 24, 512, 368, 16, 44)]
for name, B, ci, co, H, W in shapes1:
    x = torch.randn(B, ci, H, W, device=d).contiguous(memory_format=torch.channels_last)
    w = torch.randn(co, ci, 1, 1, device=d) * 0.03
    w2 = w.reshape(co, ci).contiguous()
    fl = 2 * B * H * W * co * ci
    by = B * H * W * (ci + co) * 4
    a = t(lambda: c._launch1(x, w2, co))
    b = t(lambda: F.conv2d(x, w))
    print(f"1x1 {name:32s} ours {a:8.1f} us ({fl/a/1e6:6.1f} TF, {by/a/1e6:5.2f} TB/s)   library {b:8.1f} us   x{b/a:.2f}")
