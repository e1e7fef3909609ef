"""A few dense weight-gradient launches for rocprofv3 --kernel-trace --stats (per-kernel split)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "cvpr2023-unidistill_amd")]
import torch
from unidistill_amd.ops import conv2d as c2
dev = torch.device("cuda:0"); B = 4
c2.USE_HIP_WGRAD = True
for (Cin, H, W, Cout) in [(128, 180, 180, 128), (64, 180, 180, 2688), (256, 90, 90, 256)]:
    x = torch.randn(B, Cin, H, W, device=dev, dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
    gy = torch.randn(B, Cout, H, W, device=dev, dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
    w = torch.zeros(Cout, Cin, 3, 3, device=dev)
    for _ in range(10): c2.weight_grad(x, gy, w)
    torch.cuda.synchronize()
