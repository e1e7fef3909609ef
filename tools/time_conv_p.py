"""bf16 3x3 launches of the distillation step: forward / dgrad shapes, ours (persistent 32x32x16 kernel where it applies) in
TFLOP/s; UD_CONV_P=0 in the environment times the old kernel for an A/B (run the tool twice)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "cvpr2023-unidistill_amd")]
import torch
from unidistill_amd.ops import conv2d as c2
from unidistill_amd import _lib
_lib.load().ud_conv3x3_persistent(int(os.environ.get('UD_CONV_P', '1')))
dev = torch.device("cuda:0"); B = int(os.environ.get("B", 4))
SHAPES = [("trunk b0 256->128", B, 256, 180, 180, 128), ("trunk b0 128->128", B, 128, 180, 180, 128),
          ("trunk b1 256->256", B, 256, 90, 90, 256), ("head shared 512->64", B, 512, 180, 180, 64),
          ("head c1 64->2688", B, 64, 180, 180, 2688), ("head c1 dgrad 2688->64", B, 2688, 180, 180, 64),
          ("fusion 512->256", B, 512, 180, 180, 256),
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
tot_f, tot_t = 0.0, 0.0
for name, N, Cin, H, W, Cout in SHAPES:
    x = torch.randn(N, Cin, H, W, device=dev, dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(Cout, Cin, 3, 3, device=dev) * 0.02)
    wt = c2.tap_major(w)
    flop = 2 * N * H * W * Cout * Cin * 9
    t = timeit(lambda: c2._launch(x, wt, Cout))
    ts = timeit(lambda: c2._launch(x, wt, Cout, bn_stats=True))
    tot_f += flop; tot_t += t
    print(f"{name:24s} {t:8.1f} us  {flop / t / 1e6:6.0f} TFLOP/s   with BN partials {ts:8.1f} us {flop / ts / 1e6:6.0f}", flush=True)
print(f"all shapes: {tot_f / tot_t / 1e6:.0f} TFLOP/s")
