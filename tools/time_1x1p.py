"""Persistent fp32 1x1 kernel (csrc/conv2d_f32_1x1p.hip) against the grid-per-tile kernel on the plain 1x1 shapes of one
distillation step (the table of tools/time_f32_1x1.py), with a check against an fp64 matmul:
    python tools/time_1x1p.py            all shapes
    CHECK=0 python tools/time_1x1p.py    timing only"""
import ctypes as ct
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "cvpr2023-unidistill_amd")]
os.environ.setdefault("UD_RANDOM_INIT", "1")
import torch
from unidistill_amd import _lib

dev = torch.device("cuda:0")
lib = _lib.load()
SHAPES = [  # (P, K, N, calls per step)
    (16896, 256, 1024, 11), (16896, 1024, 256, 11), (67584, 128, 512, 7), (67584, 512, 128, 7), (270336, 64, 256, 6),
    (270336, 256, 64, 5), (4224, 512, 2048, 5), (4224, 2048, 512, 5), (129600, 128, 256, 2), (270336, 64, 64, 1),
    (270336, 256, 128, 1), (67584, 512, 256, 1), (16896, 1024, 512, 1), (16896, 1024, 128, 1), (16896, 512, 368, 1),
    (129600, 256, 128, 1), (16896, 384, 512, 1), (16896, 128, 1024, 1), (16896, 512, 1024, 1), (67584, 256, 512, 1),
    (270336, 128, 256, 1), (1000, 96, 44, 0), (130, 32, 8, 0)]
CHECK = os.environ.get("CHECK", "1") == "1"
ws = torch.empty(lib.ud_conv1x1p_f32_workspace_bytes(), dtype=torch.uint8, device=dev)


def timeit(f, n=20):
    for _ in range(3):
        f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def new(x, w, y, P, K, N, bias=None, scale=None, shift=None, res=None, relu=0, part=None, use_ws=True):
    sl = ct.c_int(0)
    _lib.check(lib.ud_conv1x1p_nhwc_f32(_lib.ptr(x), _lib.ptr(w), _lib.ptr(y), P, K, N, _lib.ptr(bias), _lib.ptr(scale),
                                        _lib.ptr(shift), _lib.ptr(res), relu, _lib.ptr(part),
                                        part.numel() * 4 if part is not None else 0, ct.byref(sl), None, None, 0, 0,
                                        _lib.ptr(ws) if use_ws else None, ws.numel() if use_ws else 0, _lib.stream_of(x)),
               "ud_conv1x1p_nhwc_f32")
    return sl.value


def old(x, w, y, P, K, N):
    _lib.check(lib.ud_conv1x1_nhwc_f32(_lib.ptr(x), _lib.ptr(w), _lib.ptr(y), P, K, N, None, None, None, None, 0, _lib.stream_of(x)),
               "ud_conv1x1_nhwc_f32")


tot_o = tot_n = 0.0
print(f"{'P':>8} {'K':>5} {'N':>5} calls   old us   new us  new TF/s   no-SK us   max err (of max)")
for P, K, N, calls in SHAPES:
    torch.manual_seed(P + K + N)
    x = torch.randn(P, K, device=dev)
    w = torch.randn(N, K, device=dev) * 0.05
    y0 = torch.empty(P, N, device=dev)
    y1 = torch.empty(P, N, device=dev)
    err = float("nan")
    if CHECK:
        bias = torch.randn(N, device=dev)
        res = torch.randn(P, N, device=dev)
        part = torch.zeros(((P + 127) // 128) * N * 2, device=dev)
        y1.fill_(float("nan"))
        ns = new(x, w, y1, P, K, N, bias=bias, res=res, relu=1, part=part)
        ref = torch.relu(x.double() @ w.double().t() + bias.double() + res.double())
        err = float((y1.double() - ref).abs().max() / ref.abs().max())
        pr = part.view(ns, N, 2).double().sum(0)
        e1 = float((pr[:, 0] - ref.sum(0)).abs().max() / ref.sum(0).abs().max())
        e2 = float((pr[:, 1] - (ref * ref).sum(0)).abs().max() / (ref * ref).sum(0).abs().max())
        y2 = torch.empty_like(y1)
        new(x, w, y2, P, K, N, bias=bias, res=res, relu=1, part=part, use_ws=False)
        err_b = float((y2.double() - ref).abs().max() / ref.abs().max())
        assert err < 2e-5 and err_b < 2e-5 and e1 < 1e-4 and e2 < 1e-4, (P, K, N, err, err_b, e1, e2)
    to = timeit(lambda: old(x, w, y0, P, K, N)) if K % 32 == 0 and N % 4 == 0 and P >= 4224 else float("nan")
    tn = timeit(lambda: new(x, w, y1, P, K, N))
    tb = timeit(lambda: new(x, w, y1, P, K, N, use_ws=False))
    tot_o += to * calls
    tot_n += min(tn, tb) * calls
    print(f"{P:8d} {K:5d} {N:5d} {calls:5d} {to:8.1f} {tn:8.1f} {2.0 * P * K * N / tn / 1e6:8.1f} {tb:10.1f}   {err:.2e}")
print(f"per step: old {tot_o / 1e3:.2f} ms, new {tot_n / 1e3:.2f} ms")
