"""1x1 convolutions of the ResNet-50 image branch: MIOpen conv vs a plain GEMM (hipBLASLt via torch.matmul)
on channels-last bf16 activations, forward + backward."""
import torch, torch.nn.functional as F
dev = torch.device("cuda:0")
N = 24
SHAPES = [("l1 conv1 64->64", 64, 64, 64, 176, 1), ("l1 conv3 64->256", 64, 256, 64, 176, 3), ("l1 conv1' 256->64", 256, 64, 64, 176, 2),
          ("l2 conv1 256->128", 256, 128, 64, 176, 1), ("l2 conv3 128->512", 128, 512, 32, 88, 4), ("l2 conv1' 512->128", 512, 128, 32, 88, 3),
          ("l3 conv3 256->1024", 256, 1024, 16, 44, 6), ("l3 conv1' 1024->256", 1024, 256, 16, 44, 5),
          ("l4 conv3 512->2048", 512, 2048, 8, 22, 3), ("l4 conv1' 2048->512", 2048, 512, 8, 22, 2),
          ("depthnet 512->368", 512, 368, 16, 44, 1)]
def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
tot = [0, 0]
for name, cin, cout, H, W, cnt in SHAPES:
    x = torch.randn(N, cin, H, W, device=dev, dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    w = (torch.randn(cout, cin, 1, 1, device=dev, dtype=torch.bfloat16) * 0.05).requires_grad_(True)
    g = torch.randn(N, cout, H, W, device=dev, dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
    def conv():
        x.grad = w.grad = None
        F.conv2d(x, w).backward(g)
    def gemm():
        x.grad = w.grad = None
        y = (x.permute(0, 2, 3, 1).reshape(-1, cin) @ w.view(cout, cin).t()).view(N, H, W, cout).permute(0, 3, 1, 2)
        y.backward(g)
    tc, tg = timeit(conv), timeit(gemm)
    tot[0] += tc * cnt; tot[1] += tg * cnt
    print(f"{name:22s} conv fwd+bwd {tc:7.1f} us   gemm fwd+bwd {tg:7.1f} us   x{tc/tg:4.2f}   (x{cnt}/step)")
print(f"per step: conv {tot[0]/1e3:.2f} ms, gemm {tot[1]/1e3:.2f} ms")
