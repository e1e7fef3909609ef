"""Second head layer as task-grouped convs: groups x (2688/groups -> pad) vs dense block-diagonal."""
import torch, torch.nn.functional as F
dev = torch.device("cuda:0"); B = 4
def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
x = torch.randn(B, 2688, 180, 180, device=dev, dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last).requires_grad_(True)
for groups, cpg in [(1, 128), (2, 64), (3, 64), (6, 64), (6, 32), (7, 64), (14, 64), (21, 64), (42, 64)]:
    w = (torch.randn(groups * cpg, 2688 // groups, 3, 3, device=dev, dtype=torch.bfloat16) * 0.02).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    try:
        y = F.conv2d(x, w, None, 1, 1, 1, groups); g = torch.randn_like(y)
        tf = timeit(lambda: F.conv2d(x, w, None, 1, 1, 1, groups))
        def fb():
            x.grad = w.grad = None
            F.conv2d(x, w, None, 1, 1, 1, groups).backward(g)
        tfb = timeit(fb)
        print(f"groups={groups:2d} cout/group={cpg:3d}: fwd {tf*1e3:6.0f} us  bwd {1e3*(tfb-tf):6.0f} us")
    except Exception as e:
        print(groups, cpg, "failed", str(e)[:80])
