"""GPU: the persistent stream-K fp32 1x1 kernel (ud_conv1x1p_nhwc_f32, csrc/conv2d_f32_1x1p.hip) -- plain launches against an
fp64 matmul with every epilogue input (bias, folded BN, residual, ReLU, BatchNorm partial sums), ragged pixel / channel counts,
every schedule (whole units, cost-model stream-K, forced stream-K); mapped launches (strided, transposed, im2col: forward and
data gradient of the reference's strided blocks, base_bev_backbone.py:38-115) against the grid-per-tile kernel and the library.
fp32 products and accumulation everywhere: 2e-5 of the output's max."""
import ctypes as ct

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _cl(t):
    return t.contiguous(memory_format=torch.channels_last)


def _run(lib, _lib, x, w, y, P, K, N, bias=None, scale=None, shift=None, res=None, relu=0, part=None, ws=None, imap=None,
         omap=None):
    ns = ct.c_int(0)
    _lib.check(lib.ud_conv1x1p_nhwc_f32(_lib.ptr(x), _lib.ptr(w), _lib.ptr(y), P, K, N, _lib.ptr(bias), _lib.ptr(scale),
                                        _lib.ptr(shift), _lib.ptr(res), relu, _lib.ptr(part),
                                        part.numel() * 4 if part is not None else 0, ct.byref(ns), imap, omap, x.numel(),
                                        y.numel(), _lib.ptr(ws), ws.numel() if ws is not None else 0, _lib.stream_of(x)),
               "ud_conv1x1p_nhwc_f32")
    return ns.value


# (P, K, N): whole rounds + tails, fewer units than workgroups, one-slice units, pixel / channel counts off the tile sizes
SHAPES = [(16896, 1024, 256), (4224, 2048, 512), (16896, 256, 1024), (67584, 512, 128), (70000, 64, 64), (1000, 96, 44),
          (130, 32, 8), (129, 2048, 68), (33000, 32, 200), (5000, 640, 320)]


@pytest.mark.parametrize("shape", SHAPES)
@pytest.mark.parametrize("sk", [-1, 0, 2])
def test_persistent_1x1_plain_vs_fp64(hip_lib, shape, sk):
    from unidistill_amd import _lib
    lib = _lib.load()
    P, K, N = shape
    torch.manual_seed(P + K + N)
    dev = torch.device("cuda")
    x = torch.randn(P, K, device=dev)
    w = torch.randn(N, K, device=dev) * 0.05
    bias, scale, shift = torch.randn(N, device=dev), torch.rand(N, device=dev) + 0.5, torch.randn(N, device=dev)
    res = torch.randn(P, N, device=dev)
    ws = torch.empty(lib.ud_conv1x1p_f32_workspace_bytes(), dtype=torch.uint8, device=dev)
    part = torch.full((((P + 127) // 128) * N * 2,), float("nan"), device=dev)
    lib.ud_conv1x1p_stream_k(sk)
    try:
        y = torch.full((P, N), float("nan"), device=dev)
        ns = _run(lib, _lib, x, w, y, P, K, N, bias=bias, scale=scale, shift=shift, res=res, relu=1, part=part, ws=ws)
        y2 = torch.full((P, N), float("nan"), device=dev)
        part2 = torch.full_like(part, float("nan"))
        _run(lib, _lib, x, w, y2, P, K, N, bias=bias, scale=scale, shift=shift, res=res, relu=1, part=part2, ws=ws)
        plain = torch.full((P, N), float("nan"), device=dev)
        _run(lib, _lib, x, w, plain, P, K, N, ws=ws)
    finally:
        lib.ud_conv1x1p_stream_k(-1)
    assert torch.equal(y, y2) and torch.equal(part, part2), "not reproducible"
    lin = x.double() @ w.double().t()
    ref = torch.relu((lin + bias.double()) * scale.double() + shift.double() + res.double())
    assert ns == (P + 127) // 128
    err = float((y.double() - ref).abs().max() / ref.abs().max())
    err0 = float((plain.double() - lin).abs().max() / lin.abs().max())
    assert err <= 2e-5 and err0 <= 2e-5, (shape, sk, err, err0)
    pr = part.view(ns, N, 2).double().sum(0)
    s1, s2 = ref.sum(0), (ref * ref).sum(0)
    assert float((pr[:, 0] - s1).abs().max()) <= 1e-4 * float(s1.abs().max()) + 1e-6
    assert float((pr[:, 1] - s2).abs().max()) <= 1e-4 * float(s2.abs().max()) + 1e-6


def test_persistent_1x1_without_workspace_and_argument_checks(hip_lib):
    from unidistill_amd import _lib
    lib = _lib.load()
    dev = torch.device("cuda")
    P, K, N = 16896, 1024, 256
    x = torch.randn(P, K, device=dev)
    w = torch.randn(N, K, device=dev) * 0.05
    y = torch.full((P, N), float("nan"), device=dev)
    _run(lib, _lib, x, w, y, P, K, N)                       # no workspace: whole units only
    ref = x.double() @ w.double().t()
    assert float((y.double() - ref).abs().max() / ref.abs().max()) <= 2e-5
    ns = ct.c_int(0)
    args = [_lib.ptr(x), _lib.ptr(w), _lib.ptr(y), P, 48, N] + [None] * 4 + [0, None, 0, ct.byref(ns), None, None, 0, 0, None, 0,
                                                                          _lib.stream_of(x)]
    assert lib.ud_conv1x1p_nhwc_f32(*args) != 0             # Cin % 32 != 0
    args[4], args[6], args[7] = K, None, _lib.ptr(w)
    assert lib.ud_conv1x1p_nhwc_f32(*args) != 0             # scale without shift


@pytest.mark.parametrize("kind,cfg", [("patch", (2, 256, 128, 4, 16, 44)), ("patch", (1, 64, 64, 2, 10, 18)),
                                      ("tpatch", (2, 256, 256, 2, 12, 20)), ("tpatch", (1, 64, 64, 2, 5, 9)),
                                      ("s1x1", (2, 256, 512, 2, 16, 44)), ("s1x1", (1, 64, 128, 2, 9, 7)),
                                      ("s3x3", (2, 128, 128, 2, 32, 88)), ("s3x3", (1, 128, 256, 2, 45, 45)),
                                      ("s3x3", (2, 64, 64, 2, 9, 7)), ("s3x3", (4, 128, 256, 2, 180, 180))])
@pytest.mark.parametrize("sk", [-1, 2])
def test_persistent_1x1_mapped_convs_vs_library(hip_lib, monkeypatch, kind, cfg, sk):
    """The strided / transposed blocks routed to the persistent kernel whatever their reduction length: forward and data
    gradient equal the library's fp32 convolutions up to summation order, and the launches really went there."""
    from unidistill_amd import _lib
    from unidistill_amd.ops import conv2d as c, conv2d_f32 as c32
    lib = _lib.load()
    monkeypatch.setattr(c32, "P1X1_MIN_K_MAPPED", 32)
    monkeypatch.setattr(c32, "P1X1_KEEP_ONE_ROUND", False)
    calls = []
    real = c32.launch_1x1p
    monkeypatch.setattr(c32, "launch_1x1p", lambda *a, **k: (calls.append(1), real(*a, **k))[1])
    B, cin, cout, s, H, W = cfg
    torch.manual_seed(sum(cfg))
    x = _cl(torch.randn(B, cin, H, W, device="cuda")).requires_grad_(True)
    lib.ud_conv1x1p_stream_k(sk)
    try:
        if kind == "tpatch":
            w = (torch.randn(cin, cout, s, s, device="cuda") * 0.05).requires_grad_(True)
            ref = F.conv_transpose2d(x, w, None, s)
            y = c.conv_transpose_patch(x, w, s)
        elif kind == "patch":
            w = (torch.randn(cout, cin, s, s, device="cuda") * 0.05).requires_grad_(True)
            ref = F.conv2d(x, w, None, s)
            y = c.conv_patch(x, w, s)
        elif kind == "s1x1":
            w = (torch.randn(cout, cin, 1, 1, device="cuda") * 0.05).requires_grad_(True)
            ref = F.conv2d(x, w, None, s)
            y = c.conv1x1_strided(x, w, s)
        else:
            w = (torch.randn(cout, cin, 3, 3, device="cuda") * 0.05).requires_grad_(True)
            ref = F.conv2d(x, w, None, 2, 1)
            y = c.conv3x3_stride2(x, w)
        gy = _cl(torch.randn_like(ref))
        gx_ref, = torch.autograd.grad(ref, (x,), gy)
        gx, = torch.autograd.grad(y, (x,), gy)
    finally:
        lib.ud_conv1x1p_stream_k(-1)
    assert len(calls) >= 2, "the persistent kernel was not used"
    for a, b, name in ((y, ref, "y"), (gx, gx_ref, "dx")):
        np.testing.assert_allclose(a.detach().cpu().numpy(), b.detach().cpu().numpy(), rtol=0,
                                   atol=3e-5 * float(b.detach().abs().max()) + 1e-7, err_msg=f"{kind} {cfg} {name}")


def test_persistent_1x1_routing_and_switch(hip_lib):
    """conv1x1 (forward with BatchNorm sums, data gradient with the skip gradient added in the kernel) through the routing of
    ops/conv2d_f32.py: the same numbers with the persistent kernel on and off."""
    from unidistill_amd import _lib
    from unidistill_amd.ops import conv2d_f32 as c32
    lib = _lib.load()
    torch.manual_seed(5)
    x = _cl(torch.randn(6, 512, 16, 44, device="cuda")).requires_grad_(True)
    w = (torch.randn(128, 512, 1, 1, device="cuda") * 0.05).requires_grad_(True)
    outs = []
    for mode in (1, 0):
        lib.ud_conv1x1_f32_persistent(mode)
        try:
            assert c32.persistent_1x1(512) == (mode == 1)
            y = c32.conv1x1(x, w)
            gx, gw = torch.autograd.grad(y, (x, w), torch.ones_like(y))
            outs.append((y.detach(), gx, gw))
        finally:
            lib.ud_conv1x1_f32_persistent(-1)
    for a, b in zip(*outs):
        assert float((a - b).abs().max()) <= 2e-5 * float(b.abs().max())
