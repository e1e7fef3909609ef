"""Dense 3x3 weight gradient: hand-written kernels vs fp32 reference on ragged shapes (GPU)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "cvpr2023-unidistill_amd")]
import torch, torch.nn.functional as F
from unidistill_amd.ops import conv2d as c2
dev = torch.device("cuda:0")
torch.manual_seed(0)
bad = 0
for (B, Cin, H, W, Cout) in [(2, 64, 20, 20, 64), (1, 128, 37, 53, 72), (3, 64, 8, 16, 64), (2, 192, 33, 17, 136),
                             (1, 64, 1, 1, 8), (4, 64, 100, 90, 128)]:
    x = torch.randn(B, Cin, H, W, device=dev).bfloat16().contiguous(memory_format=torch.channels_last)
    gy = torch.randn(B, Cout, H, W, device=dev).bfloat16().contiguous(memory_format=torch.channels_last)
    w = torch.zeros(Cout, Cin, 3, 3, device=dev, requires_grad=True)
    F.conv2d(x.float(), w, None, 1, 1).backward(gy.float())
    ref = w.grad
    c2.USE_HIP_WGRAD = True
    got = c2.weight_grad(x, gy, w.detach())
    again = c2.weight_grad(x, gy, w.detach())
    rel = ((got - ref).norm() / ref.norm()).item()
    mx = ((got - ref).abs().max() / ref.abs().max()).item()
    ok = rel < 2e-3 and torch.equal(got, again)
    bad += not ok
    print(f"B{B} {Cin}->{Cout} {H}x{W}: rel {rel:.2e} max {mx:.2e} deterministic {torch.equal(got, again)} {'OK' if ok else 'FAIL'}")
print("FAILED" if bad else "ALL OK")
