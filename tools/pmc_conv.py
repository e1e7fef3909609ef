"""One conv3x3 bf16 shape, ours only, a few launches: the target of the SQ counter passes in tools/pmc_conv.sh."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "cvpr2023-unidistill_amd")]
import torch
from unidistill_amd.ops import conv2d as c2
dev = torch.device("cuda:0")
N, Cin, H, W, Cout = [int(v) for v in os.environ.get("SHAPE", "4,128,180,180,128").split(",")]
x = torch.randn(N, Cin, H, W, device=dev, dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
wt = c2.tap_major(torch.randn(Cout, Cin, 3, 3, device=dev) * 0.02)
for _ in range(10): c2._launch(x, wt, Cout)
torch.cuda.synchronize()
