"""Bitwise run-to-run determinism of the hand-written kernels (same inputs, repeated launches)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "cvpr2023-unidistill_amd")]
import torch, torch.nn.functional as F
from unidistill_amd.ops import conv2d as c2, bn_act, head_tail as ht
from unidistill_amd.layers.dense import batchnorm_act
dev = torch.device("cuda:0")
torch.manual_seed(0)
def check(name, fn, n=6):
    ref = fn()
    ref = [r.clone() for r in (ref if isinstance(ref, (list, tuple)) else [ref])]
    bad = 0
    for _ in range(n):
        out = fn(); out = out if isinstance(out, (list, tuple)) else [out]
        torch.cuda.synchronize()
        bad += sum(int(not torch.equal(a, b)) for a, b in zip(out, ref))
    print(f"{name:40s} {'DETERMINISTIC' if bad == 0 else 'DIFFERS x%d' % bad}")
for (N, Cin, H, W, Cout) in [(4, 128, 180, 180, 128), (4, 64, 180, 180, 2688), (4, 512, 180, 180, 64), (24, 256, 16, 44, 256)]:
    x = torch.randn(N, Cin, H, W, device=dev).bfloat16().contiguous(memory_format=torch.channels_last)
    w = torch.randn(Cout, Cin, 3, 3, device=dev) * 0.03
    wt = c2.tap_major(w)
    check(f"conv3x3 fwd {Cin}->{Cout} @{H}x{W}", lambda: c2._launch(x, wt, Cout))
    gy = torch.randn(N, Cout, H, W, device=dev).bfloat16().contiguous(memory_format=torch.channels_last)
    c2.USE_HIP_WGRAD = True
    check(f"conv3x3 wgrad {Cin}->{Cout}", lambda: c2.weight_grad(x, gy, w))
    c2.USE_HIP_WGRAD = False
    check(f"MIOpen wgrad {Cin}->{Cout}", lambda: c2.weight_grad(x, gy, w))
    wb = w.bfloat16().contiguous(memory_format=torch.channels_last)
    check(f"MIOpen conv fwd {Cin}->{Cout}", lambda: F.conv2d(x, wb, None, 1, 1))
c2.USE_HIP_WGRAD = "auto"
bn = torch.nn.BatchNorm2d(256).to(dev).train()
x = torch.randn(24, 256, 64, 176, device=dev).bfloat16().contiguous(memory_format=torch.channels_last)
gy = torch.randn_like(x)
def bn_step():
    xs = x.clone().requires_grad_(True); bn.zero_grad()
    y = batchnorm_act(bn, xs); y.backward(gy)
    return [y.detach(), xs.grad, bn.weight.grad.clone(), bn.bias.grad.clone()]
check("bn_act fwd+bwd 256ch", bn_step)
G = 42
y = torch.randn(4, G * 64, 180, 180, device=dev).bfloat16().contiguous(memory_format=torch.channels_last)
gam = torch.ones(G * 64, device=dev, requires_grad=True); bet = torch.zeros(G * 64, device=dev, requires_grad=True)
w2 = (torch.randn(G * 3, 64, 3, 3, device=dev) * 0.05).requires_grad_(True); b2 = torch.zeros(G * 3, device=dev, requires_grad=True)
gz = torch.randn(4, G * 3, 180, 180, device=dev)
def tail_step():
    ys = y.clone().requires_grad_(True)
    for p in (gam, bet, w2, b2): p.grad = None
    z = ht.head_tail(ys, gam, bet, w2, b2, None, None, True, 0.1, 1e-5, G, 3); z.backward(gz)
    return [z.detach(), ys.grad, w2.grad.clone(), gam.grad.clone()]
check("head_tail fwd+bwd", tail_step, n=3)
