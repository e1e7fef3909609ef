import torch, time
d = torch.device("cuda:0")
B = 4
x = torch.randn(B, 2688, 180, 180, device=d, dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last).requires_grad_(True)
for groups, cout in ((42, 126), (1, 126), (1, 128)):
    w = torch.randn(cout, 2688 // groups, 3, 3, device=d, dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    def run():
        y = torch.nn.functional.conv2d(x, w, None, padding=1, groups=groups)
        y.sum().backward()
    for _ in range(3): run()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): run()
    torch.cuda.synchronize()
    print(f"groups={groups} cout={cout}: {(time.perf_counter()-t0)/5*1e3:.2f} ms fwd+bwd", flush=True)
# 42 separate small convs on channel slices
ws = [torch.randn(3, 64, 3, 3, device=d, dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last).requires_grad_(True) for _ in range(42)]
def run2():
    ys = [torch.nn.functional.conv2d(x[:, g*64:(g+1)*64], ws[g], None, padding=1) for g in range(42)]
    torch.cat(ys, 1).sum().backward()
for _ in range(3): run2()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): run2()
torch.cuda.synchronize()
print(f"42 sliced convs: {(time.perf_counter()-t0)/5*1e3:.2f} ms fwd+bwd", flush=True)
