"""Permutation / strided convolutions on the mapped 1x1 MFMA kernels vs the library (bf16, channels-last), forward and
forward + backward, at the shapes of the distillation step."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "cvpr2023-unidistill_amd")]
import torch, torch.nn.functional as F
from unidistill_amd.ops import conv2d as c
d = torch.device("cuda:0")
def t(fn, n=10):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
cases = [("neck conv k4s4 256->128 @64x176 x24", "patch", 24, 256, 128, 4, 64, 176),
         ("neck conv k2s2 512->128 @32x88 x24", "patch", 24, 512, 128, 2, 32, 88),
         ("neck convT k2s2 2048->128 @8x22 x24", "tpatch", 24, 2048, 128, 2, 8, 22),
         ("trunk convT k2s2 256->256 @90^2 x4", "tpatch", 4, 256, 256, 2, 90, 90),
         ("shortcut 1x1 s2 256->512 @64x176 x24", "s1x1", 24, 256, 512, 2, 64, 176),
         ("shortcut 1x1 s2 512->1024 @32x88 x24", "s1x1", 24, 512, 1024, 2, 32, 88),
         ("shortcut 1x1 s2 1024->2048 @16x44 x24", "s1x1", 24, 1024, 2048, 2, 16, 44),
         ("3x3 s2 128->128 @64x176 x24", "s3x3", 24, 128, 128, 2, 64, 176),
         ("3x3 s2 256->256 @32x88 x24", "s3x3", 24, 256, 256, 2, 32, 88),
         ("3x3 s2 512->512 @16x44 x24", "s3x3", 24, 512, 512, 2, 16, 44),
         ("trunk 3x3 s2 128->256 @180^2 x4", "s3x3", 4, 128, 256, 2, 180, 180)]
for name, kind, B, ci, co, s, H, W in cases:
    x = torch.randn(B, ci, H, W, device=d).bfloat16().contiguous(memory_format=torch.channels_last).requires_grad_(True)
    if kind == "tpatch":
        w = (torch.randn(ci, co, s, s, device=d) * 0.03).requires_grad_(True)
        ours = lambda: c.conv_transpose_patch(x, w, s)
        lib = lambda: F.conv_transpose2d(x, w.to(torch.bfloat16), None, stride=s)
    elif kind == "patch":
        w = (torch.randn(co, ci, s, s, device=d) * 0.03).requires_grad_(True)
        ours = lambda: c.conv_patch(x, w, s)
        lib = lambda: F.conv2d(x, w.to(torch.bfloat16).contiguous(memory_format=torch.channels_last), None, stride=s)
    elif kind == "s1x1":
        w = (torch.randn(co, ci, 1, 1, device=d) * 0.03).requires_grad_(True)
        ours = lambda: c.conv1x1_strided(x, w, s)
        lib = lambda: F.conv2d(x, w.to(torch.bfloat16), None, stride=s)
    else:
        w = (torch.randn(co, ci, 3, 3, device=d) * 0.03).requires_grad_(True)
        ours = lambda: c.conv3x3_stride2(x, w)
        lib = lambda: F.conv2d(x, w.to(torch.bfloat16).contiguous(memory_format=torch.channels_last), None, stride=2, padding=1)
    y = ours(); gy = torch.randn_like(y)
    def fb(f):
        def run():
            x.grad = None; w.grad = None
            f().backward(gy)
        return run
    with torch.no_grad():
        a, b = t(ours), t(lib)
    a2, b2 = t(fb(ours)), t(fb(lib))
    print(f"{name:42s} fwd ours {a:7.1f} us  lib {b:7.1f} us  x{b/a:4.2f} | fwd+bwd ours {a2:7.1f} us  lib {b2:7.1f} us  x{b2/a2:4.2f}")
