"""Hand-written MFMA conv3x3 vs MIOpen on the dense conv shapes of the distillation step (bf16 NHWC)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "cvpr2023-unidistill_amd")]
import torch, torch.nn.functional as F
from unidistill_amd.ops import conv2d as c2
dev = torch.device("cuda:0"); B = int(os.environ.get("B", 4))
SHAPES = [("trunk b0 256->128", B, 256, 180, 180, 128), ("trunk b0 128->128", B, 128, 180, 180, 128),
          ("trunk b1 256->256", B, 256, 90, 90, 256), ("head shared 512->64", B, 512, 180, 180, 64),
          ("head c1 64->2688", B, 64, 180, 180, 2688), ("head c1 dgrad 2688->64", B, 2688, 180, 180, 64), ("lss depth 512->512", 6 * B, 512, 16, 44, 512),
          ("resnet l1 64->64", 6 * B, 64, 64, 176, 64), ("resnet l2 128->128", 6 * B, 128, 32, 88, 128),
          ("resnet l3 256->256", 6 * B, 256, 16, 44, 256), ("resnet l4 512->512", 6 * B, 512, 8, 22, 512)]
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for name, N, Cin, H, W, Cout in ([] if os.environ.get("ONLY_WGRAD") else SHAPES):
    x = torch.randn(N, Cin, H, W, device=dev, dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(Cout, Cin, 3, 3, device=dev) * 0.02)
    wb = w.bfloat16().contiguous(memory_format=torch.channels_last)
    wt = c2.tap_major(w)
    flop = 2 * N * H * W * Cout * Cin * 9
    t_lib = timeit(lambda: F.conv2d(x, wb, None, 1, 1))
    t_our = timeit(lambda: c2._launch(x, wt, Cout))
    err = (c2._launch(x, wt, Cout).float() - F.conv2d(x, wb, None, 1, 1).float()).abs().max().item()
    print(f"{name:22s} MIOpen {t_lib:7.1f} us ({flop/t_lib/1e6:5.0f} TF)   ours {t_our:7.1f} us ({flop/t_our/1e6:5.0f} TF)   x{t_lib/t_our:4.2f}  maxdiff {err:.3g}")
print("--- weight gradient ---")
for name, N, Cin, H, W, Cout in SHAPES:
    x = torch.randn(N, Cin, H, W, device=dev, dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
    gy = torch.randn(N, Cout, H, W, device=dev, dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(Cout, Cin, 3, 3, device=dev) * 0.02)
    flop = 2 * N * H * W * Cout * Cin * 9
    c2.USE_HIP_WGRAD = False; ref = c2.weight_grad(x, gy, w)
    t_lib = float("nan") if os.environ.get("NO_LIB") else timeit(lambda: c2.weight_grad(x, gy, w))
    c2.USE_HIP_WGRAD = True; t_our = timeit(lambda: c2.weight_grad(x, gy, w)); got = c2.weight_grad(x, gy, w)
    err = ((got - ref).norm() / ref.norm()).item()
    print(f"{name:22s} MIOpen {t_lib:7.1f} us ({flop/t_lib/1e6:5.0f} TF)   ours {t_our:7.1f} us ({flop/t_our/1e6:5.0f} TF)   x{t_lib/t_our:4.2f}  rel diff {err:.3g}")
