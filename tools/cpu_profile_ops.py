"""cProfile of the Python op wrappers (tiny tensors): where the host time per call goes."""
import os, sys, cProfile, pstats
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "cvpr2023-unidistill_amd")]
import torch
from torch import nn
from unidistill_amd.layers import dense
dev = torch.device("cuda:0")
x = torch.randn(1, 64, 8, 16, device=dev).bfloat16().contiguous(memory_format=torch.channels_last)
xg = x.clone().requires_grad_(True)
conv3 = dense.Conv2d(64, 64, 3, padding=1, bias=False).to(dev)
bn = nn.BatchNorm2d(64).to(dev).train()
which = sys.argv[1] if len(sys.argv) > 1 else "bn"
def body():
    with torch.autocast("cuda", dtype=torch.bfloat16):
        for _ in range(1000):
            if which == "bn":
                y = dense.batchnorm_act(bn, xg)
            else:
                y = conv3(xg)
            y.backward(y)
body()
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable(); body(); pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
