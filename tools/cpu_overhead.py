"""Host-side cost per call of the Python op wrappers (tiny tensors: the GPU is never the limit)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "cvpr2023-unidistill_amd")]
import torch, torch.nn.functional as F
from torch import nn
from unidistill_amd.ops import conv2d as c2, bn_act as hb
from unidistill_amd.layers import dense
dev = torch.device("cuda:0")
def bench(name, fn, n=400):
    for _ in range(20): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    t1 = time.perf_counter(); torch.cuda.synchronize()
    print(f"{name:44s} {(t1 - t0) / n * 1e6:7.1f} us/call (host)")
x = torch.randn(1, 64, 8, 16, device=dev).bfloat16().contiguous(memory_format=torch.channels_last)
xg = x.clone().requires_grad_(True)
conv3 = dense.Conv2d(64, 64, 3, padding=1, bias=False).to(dev)
conv1 = dense.Conv2d(64, 64, 1, bias=False).to(dev)
bn = nn.BatchNorm2d(64).to(dev).train()
lib3 = nn.Conv2d(64, 64, 3, padding=1, bias=False).to(dev)
with torch.autocast("cuda", dtype=torch.bfloat16):
    bench("nn.Conv2d 3x3 fwd (library, autocast)", lambda: lib3(xg))
    bench("dense.Conv2d 3x3 fwd (ours)", lambda: conv3(xg))
    bench("dense.Conv2d 1x1 fwd (ours, gemm path)", lambda: conv1(xg))
    bench("nn.BatchNorm2d+relu fwd (library)", lambda: F.relu(bn(xg)))
    bench("batchnorm_act fwd (ours)", lambda: dense.batchnorm_act(bn, xg))
    def fb3():
        y = conv3(xg); y.backward(y)
    def fbl():
        y = lib3(xg); y.backward(y)
    def fbb():
        y = dense.batchnorm_act(bn, xg); y.backward(y)
    def fbbl():
        y = F.relu(bn(xg)); y.backward(y)
    bench("nn.Conv2d 3x3 fwd+bwd (library)", fbl)
    bench("dense.Conv2d 3x3 fwd+bwd (ours)", fb3)
    bench("nn.BatchNorm2d+relu fwd+bwd (library)", fbbl)
    bench("batchnorm_act fwd+bwd (ours)", fbb)
