"""Frozen ResNet stem at the benchmark shape (24 images 256 x 704): csrc/stem.hip vs the library path (MIOpen convolution +
BatchNorm / ReLU + max-pool kernels)."""
import os as _os; _os.environ.setdefault("UD_RANDOM_INIT", "1")
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "cvpr2023-unidistill_amd")]
import torch
import torch.nn.functional as F
from unidistill_amd.ops import stem
from unidistill_amd.layers.dense import batchnorm_act

d = torch.device("cuda:0")
conv = torch.nn.Conv2d(3, 64, 7, stride=2, padding=3, bias=False).to(d)
bn = torch.nn.BatchNorm2d(64).to(d).eval()
for p in list(conv.parameters()) + list(bn.parameters()):
    p.requires_grad = False
conv.weight.data = conv.weight.data.contiguous(memory_format=torch.channels_last)
pool = torch.nn.MaxPool2d(3, 2, 1)
x = torch.randn(24, 3, 256, 704, device=d)


def timed(fn, n=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def library():
    with torch.no_grad():
        xc = x.contiguous(memory_format=torch.channels_last)
        return pool(batchnorm_act(bn, conv(xc)))


flops = 2 * 24 * 128 * 352 * 64 * 147
for name, fn in (("hip stem f32", lambda: stem.stem(x, conv, bn)),
                 ("hip conv+bn+relu only", lambda: stem.stem(x, conv, bn, pool=False)),
                 ("hip stem bf16 out", lambda: stem.stem(x, conv, bn, torch.bfloat16)),
                 ("library path f32", library)):
    us = timed(fn)
    print(f"{name:24s} {us:8.1f} us   ({flops / us / 1e6:6.1f} TFLOP/s of the 7x7 convolution's direct flops)")
