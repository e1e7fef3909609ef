"""1x1 convolution forward / data gradient on the ResNet-50 bottleneck shapes: MIOpen vs GEMM vs ud_conv1x1."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "cvpr2023-unidistill_amd")]
import torch, torch.nn.functional as F
from unidistill_amd.ops import conv2d as c2
dev = torch.device("cuda:0"); N = 6 * int(os.environ.get("B", 4))
SHAPES = [("l1 64->64", 64, 64, 176, 64), ("l1 64->256", 64, 64, 176, 256), ("l1 256->64", 256, 64, 176, 64), ("l2 256->128", 256, 64, 176, 128),
          ("l2 128->512", 128, 32, 88, 512), ("l2 512->128", 512, 32, 88, 128), ("l3 512->256", 512, 32, 88, 256),
          ("l3 256->1024", 256, 16, 44, 1024), ("l3 1024->256", 1024, 16, 44, 256),
          ("l4 512->2048", 512, 8, 22, 2048), ("l4 2048->512", 2048, 8, 22, 512), ("depth 512->368", 512, 16, 44, 368)]
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for name, Cin, H, W, Cout in SHAPES:
    x = torch.randn(N, Cin, H, W, device=dev, dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
    w = torch.randn(Cout, Cin, 1, 1, device=dev) * Cin ** -0.5
    wb = w.bfloat16().contiguous(memory_format=torch.channels_last)
    w2 = wb.view(Cout, Cin)
    P = N * H * W
    x2 = x.permute(0, 2, 3, 1).reshape(P, Cin)
    t_conv = timeit(lambda: F.conv2d(x, wb))
    t_gemm = timeit(lambda: x2 @ w2.t())
    t_our = timeit(lambda: c2._launch1x1(x, w2, Cout))
    err = (c2._launch1x1(x, w2, Cout).float() - F.conv2d(x.float(), w.bfloat16().float())).abs().max().item()
    mb = (x.numel() + P * Cout) * 2 / 1e6
    print(f"{name:16s} {mb:6.0f} MB  MIOpen {t_conv:7.1f} us  GEMM {t_gemm:7.1f} us  ours {t_our:7.1f} us ({mb/t_our:5.2f} TB/s, {2*P*Cin*Cout/t_our/1e6:4.0f} TF)  max err {err:.2e}")
