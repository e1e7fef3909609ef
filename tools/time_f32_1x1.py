"""Every plain fp32 1x1-kernel launch (forward + data gradient) of one distillation step, timed per distinct shape.  The tile-width choice is read once per process:
    python tools/time_f32_1x1.py            (the launcher's own choice)
    UD_F32_TN64=1 python tools/time_f32_1x1.py   (64-wide output-channel tiles everywhere)"""
import collections
import ctypes as ct
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "cvpr2023-unidistill_amd")]
os.environ.setdefault("UD_RANDOM_INIT", "1")
import torch
from unidistill_amd import train as T
from unidistill_amd.ops import conv2d as c2, conv2d_f32 as cf

dev = torch.device("cuda:0")
torch.manual_seed(0)
trainer = T.Trainer(T.DistillStep("camera_exp_distill_lidar"), device=dev, autocast_dtype=None, channels_last=True)
batch = T.synthetic_batch(dev, 4)
for _ in range(2):
    trainer.step(batch)
cf.LOG_1X1 = []
trainer.step(batch)
torch.cuda.synchronize()
log, cf.LOG_1X1 = cf.LOG_1X1, None
del trainer
torch.cuda.empty_cache()
cnt = collections.Counter((e[0], e[1], e[2], e[3], e[4], e[5]) + tuple(e[6:]) for e in log)


def timeit(f):
    for _ in range(3):
        f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 10


tot = 0.0
print(f"{'kind':>6} {'P':>8} {'K':>5} {'N':>5}  calls   ms   TFLOP/s  TB/s  ms/step  maps")
for key, n in sorted(cnt.items(), key=lambda kv: -kv[1]):
    kind, P, K, N, imap, omap = key[:6]
    if kind == "line":
        x = torch.randn(1, K, P // 16, 16, device=dev).contiguous(memory_format=torch.channels_last) if P % 16 == 0 else \
            torch.randn(1, K, P, 1, device=dev).contiguous(memory_format=torch.channels_last)
        w = torch.randn(N, K, device=dev) * 0.05
        ms = timeit(lambda: cf._launch1(x, w, N))
    elif os.environ.get("MAPPED") != "1":
        continue                 # pixel-mapped launches: MAPPED=1 (needs the producing layer's exact buffers; off by default)
    else:
        xs, ys = key[6], key[7]
        x = torch.randn(xs, device=dev).contiguous(memory_format=torch.channels_last)
        y = torch.empty(ys, device=dev).contiguous(memory_format=torch.channels_last)
        w = torch.randn(N, 9 * imap[6] if (imap is not None and imap[0] == 4) else K, device=dev) * 0.05   # mode 4: [C][3][3][Cout]
        im = None if imap is None else (ct.c_int * 9)(*imap)
        om = None if omap is None else (ct.c_int * 9)(*omap)
        ms = timeit(lambda: c2._mapped(x, w, y, P, K, N, im, om))
    fl = 2.0 * P * K * N
    tot += ms * n
    print(f"{kind:>6} {P:8d} {K:5d} {N:5d} {n:6d} {ms:7.3f} {fl / ms / 1e9:7.1f} {4.0 * P * (K + N) / ms / 1e9:6.2f} {ms * n:7.3f}  {imap} {omap}")
print(f"total {tot:.2f} ms/step over {sum(cnt.values())} launches")
