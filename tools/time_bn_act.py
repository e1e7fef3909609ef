"""BatchNorm(+ReLU) streaming kernels (csrc/bn_act.hip) against their HBM bounds, at the step's big shapes."""
import os as _os; _os.environ.setdefault("UD_RANDOM_INIT", "1")   # synthetic weights (tools never train for real)
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "cvpr2023-unidistill_amd")]
import torch
from unidistill_amd import _lib
from unidistill_amd.ops import bn_act as hb
dev = torch.device("cuda:0")
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for dt in (torch.bfloat16, torch.float32):
    for name, B, C, H, W in (("trunk 128 @180^2", 4, 128, 180, 180), ("trunk 256 @90^2", 4, 256, 90, 90),
                             ("deblock 256 @180^2", 4, 256, 180, 180), ("resnet 256 @64x176 x24", 24, 256, 64, 176),
                             ("resnet 512 @32x88 x24", 24, 512, 32, 88)):
        bn = torch.nn.BatchNorm2d(C).to(dev).train()
        x = torch.randn(B, C, H, W, device=dev).to(dt).contiguous(memory_format=torch.channels_last).requires_grad_(True)
        gy = torch.randn(B, C, H, W, device=dev).to(dt).contiguous(memory_format=torch.channels_last)
        nbytes = x.numel() * x.element_size()
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=dt == torch.bfloat16):
            fwd = t(lambda: hb.bn_act(bn, x, None, True))
            y = hb.bn_act(bn, x, None, True)
            def fb():
                x.grad = None
                y = hb.bn_act(bn, x, None, True)
                y.backward(gy)
            both = t(fb)
        _lib.prof_enable(True)
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=dt == torch.bfloat16):
            for _ in range(10): fb()
        torch.cuda.synchronize(); _lib.prof_enable(False)
        parts = []
        for k in ("bn_act.stats", "bn_act.k_fwd", "bn_act.k_bwd_reduce", "bn_act.k_bwd_dx", "bn_act.bwd"):
            ms, n = _lib.prof_read(k, reset=True)
            if n: parts.append(f"{k.split('.')[1]} {ms / n * 1e3:.0f}")
        # fwd: stats read x (1) + apply read x, write y (2) = 3 passes; bwd: reduce reads dy, y|x (2) + dx reads dy, x|y, writes dx (3) = 5
        print(f"{str(dt)[6:]:8s} {name:24s} {nbytes/1e6:6.1f} MB/pass  fwd {fwd:6.1f} us = {3*nbytes/fwd/1e6:4.2f} TB/s over 3 passes   "
              f"bwd {both-fwd:6.1f} us = {5*nbytes/(both-fwd)/1e6:4.2f} TB/s over 5 passes   [{', '.join(parts)} us incl. ~6 us events]")
