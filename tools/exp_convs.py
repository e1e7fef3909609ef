"""MIOpen timings of every distinct dense conv shape in the distillation step (bf16, channels-last),
to see which shapes a hand-written MFMA kernel has to beat."""
import os, sys, torch, torch.nn.functional as F
dev = torch.device("cuda:0")
B = int(os.environ.get("B", 4))
# name, N, Cin, H, W, Cout, k, stride, pad, count per step (student + teacher)
SHAPES = [
    ("trunk b0 256->128", B, 256, 180, 180, 128, 3, 1, 1, 2),
    ("trunk b0 128->128", B, 128, 180, 180, 128, 3, 1, 1, 10),
    ("trunk b1 128->256 s2", B, 128, 180, 180, 256, 3, 2, 1, 2),
    ("trunk b1 256->256", B, 256, 90, 90, 256, 3, 1, 1, 10),
    ("head shared 512->64", B, 512, 180, 180, 64, 3, 1, 1, 2),
    ("head c1 64->2688", B, 64, 180, 180, 2688, 3, 1, 1, 2),
    ("head c2 2688->128 (block-diag dense)", B, 2688, 180, 180, 128, 3, 1, 1, 2),
    ("lss depth 512->512 @16x44", 6 * B, 512, 16, 44, 512, 3, 1, 1, 1),
    ("resnet l1 64->64 @64x176", 6 * B, 64, 64, 176, 64, 3, 1, 1, 3),
    ("resnet l2 128->128 @32x88", 6 * B, 128, 32, 88, 128, 3, 1, 1, 4),
    ("resnet l3 256->256 @16x44", 6 * B, 256, 16, 44, 256, 3, 1, 1, 6),
    ("resnet l4 512->512 @8x22", 6 * B, 512, 8, 22, 512, 3, 1, 1, 3),
]
def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
tot_f = tot_b = 0
for name, N, Cin, H, W, Cout, k, s, p, cnt in SHAPES:
    x = torch.randn(N, Cin, H, W, device=dev, dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    w = (torch.randn(Cout, Cin, k, k, device=dev, dtype=torch.bfloat16) * 0.02).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    y = F.conv2d(x, w, None, s, p)
    g = torch.randn_like(y)
    flop = 2 * y.numel() * Cin * k * k
    tf = timeit(lambda: F.conv2d(x, w, None, s, p))
    def fb():
        x.grad = w.grad = None
        F.conv2d(x, w, None, s, p).backward(g)
    tfb = timeit(fb)
    tot_f += tf * cnt; tot_b += (tfb - tf) * cnt
    print(f"{name:40s} fwd {tf*1e3:7.0f} us ({flop/tf/1e9:6.0f} TF)  bwd {1e3*(tfb-tf):7.0f} us ({2*flop/(tfb-tf)/1e9:6.0f} TF)  x{cnt}")
print(f"sum over step: fwd {tot_f:.2f} ms, bwd(student+teacher counted; teacher has none) {tot_b:.2f} ms")
