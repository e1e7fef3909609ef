"""GPU: the fp32 MFMA convolutions (ud_conv3x3_nhwc_f32 / ud_conv1x1_nhwc_f32: forward + data gradient) against
plain PyTorch fp32 convolutions of the same tensors.  fp32 products and accumulation on both sides, only the
summation order differs: tolerance 2e-5 of the output's max."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _cl(t):
    return t.contiguous(memory_format=torch.channels_last)


# the last three shapes make the per-launch tile-height choice pick 6-row tiles, 8-row tiles and the 8-row narrow tile
@pytest.mark.parametrize("shape", [(2, 64, 128, 20, 36), (1, 256, 128, 45, 45), (3, 32, 64, 9, 17),
                                   (1, 128, 76, 33, 40), (2, 512, 64, 24, 24), (1, 64, 2688, 16, 16),
                                   (1, 128, 128, 180, 180), (16, 64, 128, 8, 256), (16, 64, 64, 8, 256)])
def test_conv3x3_f32_forward_and_dgrad(hip_lib, shape):
    from unidistill_amd.ops import conv2d_f32 as c
    B, cin, cout, H, W = shape
    torch.manual_seed(sum(shape))
    x = _cl(torch.randn(B, cin, H, W, device="cuda")).requires_grad_(True)
    w = (torch.randn(cout, cin, 3, 3, device="cuda") * 0.05).requires_grad_(True)
    b = torch.randn(cout, device="cuda").requires_grad_(True)
    y = c.conv3x3(x, w, b)
    gy = _cl(torch.randn_like(y))
    y.backward(gy)
    gx, gw, gb = x.grad.clone(), w.grad.clone(), b.grad.clone()
    x.grad = w.grad = b.grad = None
    with torch.backends.cudnn.flags(enabled=False):      # reference: direct (non-library-heuristic) path
        yr = F.conv2d(x, w, b, padding=1)
        yr.backward(gy)
    assert y.is_contiguous(memory_format=torch.channels_last) and y.dtype == torch.float32
    assert (y - yr).abs().max() <= 2e-5 * yr.abs().max()
    assert (gx - x.grad).abs().max() <= 2e-5 * x.grad.abs().max()
    assert (gw - w.grad).abs().max() <= 1e-4 * w.grad.abs().max()
    assert (gb - b.grad).abs().max() <= 1e-4 * b.grad.abs().max()


@pytest.mark.parametrize("shape", [(24, 64, 256, 16, 44), (2, 256, 64, 31, 33), (1, 512, 368, 16, 44), (4, 32, 128, 20, 20)])
def test_conv1x1_f32_forward_and_dgrad(hip_lib, shape):
    from unidistill_amd.ops import conv2d_f32 as c
    B, cin, cout, H, W = shape
    torch.manual_seed(sum(shape))
    x = _cl(torch.randn(B, cin, H, W, device="cuda")).requires_grad_(True)
    w = (torch.randn(cout, cin, 1, 1, device="cuda") * 0.05).requires_grad_(True)
    y = c.conv1x1(x, w, None)
    gy = _cl(torch.randn_like(y))
    y.backward(gy)
    gx, gw = x.grad.clone(), w.grad.clone()
    x.grad = w.grad = None
    yr = F.conv2d(x, w)
    yr.backward(gy)
    assert (y - yr).abs().max() <= 2e-5 * yr.abs().max()
    assert (gx - x.grad).abs().max() <= 2e-5 * x.grad.abs().max()
    assert (gw - w.grad).abs().max() <= 1e-4 * w.grad.abs().max()


def test_conv_f32_dgrad_pads_cout_off_the_slice_width(hip_lib):
    """Cout = 368 (the depth net) / 40: the data gradient reduces over Cout in 32-channel slices, so dy and the transposed
    weights are zero-padded to the next multiple -- still the hand-written kernels (profile counter), same numbers."""
    from unidistill_amd import _lib
    from unidistill_amd.ops import conv2d_f32 as c
    for ks, (B, cin, cout, H, W) in ((1, (2, 64, 368, 9, 20)), (3, (1, 32, 40, 13, 17))):
        torch.manual_seed(cout)
        x = _cl(torch.randn(B, cin, H, W, device="cuda")).requires_grad_(True)
        w = (torch.randn(cout, cin, ks, ks, device="cuda") * 0.05).requires_grad_(True)
        name = "conv2d.k_conv3x3_f32" if ks == 3 else "conv2d.k_conv1x1_f32"
        _lib.prof_read(name, reset=True)
        _lib.prof_read("conv2d.k_conv3x3_wino_f32", reset=True)
        _lib.prof_enable(True)
        y = c.conv3x3(x, w) if ks == 3 else c.conv1x1(x, w)
        gy = _cl(torch.randn_like(y))
        gx, = torch.autograd.grad(y, x, gy)
        torch.cuda.synchronize()
        _lib.prof_enable(False)
        # forward + data gradient, on the direct or (3x3) the Winograd kernel
        assert _lib.prof_read(name)[1] + (_lib.prof_read("conv2d.k_conv3x3_wino_f32")[1] if ks == 3 else 0) == 2
        gx_ref, = torch.autograd.grad(F.conv2d(x, w, None, 1, ks // 2), x, gy)
        assert (gx - gx_ref).abs().max() <= 2e-5 * gx_ref.abs().max()


def test_fp32_trunk_routes_to_the_fp32_kernels_and_matches_library(hip_lib):
    """BaseBEVBackbone in fp32 mode: ZeroPad + unpadded conv folding, stride-1 convs on ud_conv3x3_nhwc_f32."""
    from unidistill_amd import _lib
    from unidistill_amd.layers import dense
    from unidistill_amd.layers.bev import BaseBEVBackbone
    torch.manual_seed(0)
    dense.Conv2d.hip_fp32 = "all"          # everywhere, not only where the kernel measured faster than the library
    m = BaseBEVBackbone([2, 2], [1, 2], [64, 128], [1, 2], [64, 64], 64).cuda().train()
    x = _cl(torch.randn(2, 64, 40, 40, device="cuda"))
    _lib.prof_read("conv2d.k_conv3x3_f32", reset=True)
    _lib.prof_read("conv2d.k_conv3x3_wino_f32", reset=True)
    _lib.prof_enable(True)
    y, _ = m(x)
    torch.cuda.synchronize()
    _lib.prof_enable(False)
    calls = _lib.prof_read("conv2d.k_conv3x3_f32")[1] + _lib.prof_read("conv2d.k_conv3x3_wino_f32")[1]
    assert calls == 5, calls           # 3 stride-1 convs of level 0 (incl. the ZeroPad one) + 2 of level 1 (direct or Winograd kernel)
    dense.Conv2d.hip_enabled = False
    try:
        for mod in m.modules():        # same batch statistics on the second pass: reset what BN accumulated
            if isinstance(mod, torch.nn.BatchNorm2d):
                mod.reset_running_stats()
        yr, _ = m(x)
    finally:
        dense.Conv2d.hip_enabled = True
        dense.Conv2d.hip_fp32 = "all"
    assert (y - yr).abs().max() <= 1e-4 * yr.abs().max()


@pytest.mark.gpu
@pytest.mark.parametrize("H,W", [(10, 12), (40, 41)])
def test_conv1x1_f32_skip_adds_identity_gradient_in_kernel(hip_lib, H, W):
    """conv1x1_skip (fp32): the gradient of the identity branch is added in the data-gradient epilogue: same result as the
    convolution followed by a separate residual add (ResNet bottleneck joins of the fp32 mode)."""
    from unidistill_amd.ops import conv2d_f32 as c
    torch.manual_seed(H + W)
    x0 = _cl(torch.randn(2, 64, H, W, device="cuda"))
    w0 = torch.randn(32, 64, 1, 1, device="cuda") * 0.1
    w2 = torch.randn(64, 32, 1, 1, device="cuda") * 0.1          # second conv back to the input width: y2 + identity
    outs = []
    for skip in (False, True):
        x = x0.clone().requires_grad_(True)
        w = w0.clone().requires_grad_(True)
        if skip:
            y, idt = c.conv1x1_skip(x, w)
        else:
            y, idt = c.conv1x1(x, w), x
        z = torch.nn.functional.conv2d(y, w2) + idt
        (z * z).sum().backward()
        outs.append((z.detach(), x.grad.clone(), w.grad.clone()))
    for a, b in zip(*outs):
        np.testing.assert_allclose(a.cpu().numpy(), b.cpu().numpy(), rtol=1e-5, atol=1e-5 * float(b.abs().max()))


@pytest.mark.gpu
def test_conv_f32_random_shapes_vs_library(hip_lib):
    """Seeded random shapes through the fp32 3x3 and 1x1 kernels (forward + data gradient) against the library's fp32
    convolutions: ragged maps, channel counts off the tile widths, every tile-height variant."""
    from unidistill_amd.ops import conv2d_f32 as c
    rng = np.random.default_rng(11)
    for it in range(20):
        B = int(rng.integers(1, 4))
        H, W = int(rng.integers(1, 60)), int(rng.integers(1, 60))
        cin = 32 * int(rng.integers(1, 7))
        cout = 32 * int(rng.integers(1, 9))
        ks = 3 if it % 3 else 1
        torch.manual_seed(200 + it)
        x = _cl(torch.randn(B, cin, H, W, device="cuda")).requires_grad_(True)
        w = (torch.randn(cout, cin, ks, ks, device="cuda") * (cin * ks * ks) ** -0.5).requires_grad_(True)
        ref = F.conv2d(x, w, None, 1, ks // 2)
        gy = _cl(torch.randn_like(ref))
        gx_ref, = torch.autograd.grad(ref, x, gy)
        y = c.conv3x3(x, w) if ks == 3 else c.conv1x1(x, w)
        gx, = torch.autograd.grad(y, x, gy)
        tag = f"case {it}: B={B} cin={cin} H={H} W={W} cout={cout} k={ks}"
        np.testing.assert_allclose(y.detach().cpu().numpy(), ref.detach().cpu().numpy(), rtol=0,
                                   atol=3e-5 * float(ref.detach().abs().max()) + 1e-7, err_msg=tag)
        np.testing.assert_allclose(gx.cpu().numpy(), gx_ref.cpu().numpy(), rtol=0,
                                   atol=3e-5 * float(gx_ref.abs().max()) + 1e-7, err_msg=tag)


@pytest.mark.gpu
@pytest.mark.parametrize("kind,cfg", [("patch", (2, 256, 128, 4, 16, 44)), ("patch", (1, 64, 64, 2, 10, 18)),
                                      ("tpatch", (2, 256, 256, 2, 12, 20)), ("tpatch", (1, 64, 64, 2, 5, 9)),
                                      ("s1x1", (2, 256, 512, 2, 16, 44)), ("s1x1", (1, 64, 128, 2, 9, 7)),
                                      ("s3x3", (2, 128, 128, 2, 32, 88)), ("s3x3", (1, 128, 256, 2, 45, 45)),
                                      ("s3x3", (2, 64, 64, 2, 9, 7))])
def test_fp32_strided_and_transposed_convs_on_the_mapped_kernel(hip_lib, kind, cfg):
    """fp32 mode: conv k = s / stride s, transposed conv k = s / stride s, 1x1 / stride s and 3x3 / stride 2 / pad 1 run
    forward and data gradient on ud_conv1x1_mapped_nhwc_f32 (weight gradient: library): same numbers as the library's
    fp32 convolutions up to summation order."""
    from unidistill_amd.ops import conv2d as c
    B, cin, cout, s, H, W = cfg
    torch.manual_seed(sum(cfg))
    x = _cl(torch.randn(B, cin, H, W, device="cuda")).requires_grad_(True)
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
    assert y.dtype == torch.float32 and y.shape == ref.shape
    gy = _cl(torch.randn_like(ref))
    gx_ref, gw_ref = torch.autograd.grad(ref, (x, w), gy)
    gx, gw = torch.autograd.grad(y, (x, w), gy)
    for a, b, name in ((y, ref, "y"), (gx, gx_ref, "dx"), (gw, gw_ref, "dw")):
        np.testing.assert_allclose(a.detach().cpu().numpy(), b.detach().cpu().numpy(), rtol=0,
                                   atol=5e-5 * float(b.detach().abs().max()) + 1e-7, err_msg=f"{kind} {cfg} {name}")


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(2, 64, 128, 20, 36), (1, 256, 128, 45, 45), (3, 36, 52, 9, 17), (1, 128, 76, 33, 40),
                                   (2, 512, 64, 24, 24), (1, 64, 2688, 16, 16), (1, 128, 128, 180, 180),
                                   (24, 64, 64, 8, 22), (4, 4, 8, 1, 1)])
@pytest.mark.parametrize("ks", [3, 1])
def test_f32_weight_gradient_kernels(hip_lib, shape, ks):
    """ud_conv3x3_wgrad_nhwc_f32 / ud_conv1x1_wgrad_mapped_nhwc_f32 against a float64 convolution backward of the same
    tensors (fp32 products + fp32 accumulation in the kernel: 2e-5 of the gradient's max); channel counts off the 64-wide
    tiles, ragged maps, a 1 x 1 map; bitwise reproducible (fixed-order slice reduction, no atomics)."""
    from unidistill_amd.ops import conv2d_f32 as c
    B, cin, cout, H, W = shape
    torch.manual_seed(sum(shape) + ks)
    x = _cl(torch.randn(B, cin, H, W, device="cuda"))
    gy = _cl(torch.randn(B, cout, H, W, device="cuda"))
    w = torch.zeros(cout, cin, ks, ks, device="cuda")
    assert c.wgrad_supported(x, gy)
    gw = c.weight_grad(x, gy, w, ks)
    gw2 = c.weight_grad(x, gy, w, ks)
    assert gw.shape == w.shape and torch.equal(gw, gw2)
    ref = torch.ops.aten.convolution_backward(gy.double(), x.double(), w.double(), None, [1, 1], [ks // 2] * 2, [1, 1],
                                              False, [0, 0], 1, [False, True, False])[1]
    err = (gw.double() - ref).abs().max().item()
    assert err <= 2e-5 * ref.abs().max().item() + 1e-9, (shape, ks, err, ref.abs().max().item())


@pytest.mark.gpu
def test_fp32_training_step_has_no_library_weight_gradient(hip_lib):
    """fp32 mode: every weight gradient of the BEV trunk (stride-1, ZeroPad + conv, strided level, both deblocks) comes from
    the hand-written kernels and matches the library path."""
    from unidistill_amd import _lib
    from unidistill_amd.layers import dense
    from unidistill_amd.layers.bev import BaseBEVBackbone
    torch.manual_seed(0)
    m = BaseBEVBackbone([2, 2], [1, 2], [64, 128], [1, 2], [64, 64], 64).cuda().train()
    x = _cl(torch.randn(2, 64, 40, 40, device="cuda"))
    grads = []
    for hip in (True, False):
        dense.Conv2d.hip_enabled = hip
        try:
            for mod in m.modules():
                if isinstance(mod, torch.nn.BatchNorm2d):
                    mod.reset_running_stats()
            m.zero_grad(set_to_none=True)
            for nm in ("conv2d.k_wgrad_f32", "conv2d.k_wgrad_wino_f32", "conv2d.k_wgrad_1x1_f32"):
                _lib.prof_read(nm, reset=True)
            _lib.prof_enable(hip)
            y, _ = m(x)
            y.square().mean().backward()
            torch.cuda.synchronize()
            _lib.prof_enable(False)
        finally:
            dense.Conv2d.hip_enabled = True
        if hip:
            n3 = _lib.prof_read("conv2d.k_wgrad_f32")[1] + _lib.prof_read("conv2d.k_wgrad_wino_f32")[1]   # direct or Winograd
            n1 = _lib.prof_read("conv2d.k_wgrad_1x1_f32")[1]
            assert n3 == 5 and n1 == 3, (n3, n1)       # 5 stride-1 3x3 convs; strided 3x3 + the two deblocks
        grads.append({n: p.grad.clone() for n, p in m.named_parameters() if p.dim() == 4})
    for n, g in grads[0].items():
        r = grads[1][n]
        assert (g - r).abs().max() <= 2e-4 * r.abs().max(), n


@pytest.mark.parametrize("shape", [(1, 64, 64, 16, 16), (2, 128, 128, 180, 180), (1, 256, 256, 90, 90), (3, 64, 192, 64, 176),
                                   (2, 128, 64, 32, 88), (5, 256, 256, 16, 44), (1, 512, 64, 8, 22), (2, 72, 100, 27, 35),
                                   (1, 64, 2688, 36, 28), (1, 64, 64, 1, 1), (1, 8, 4, 5, 3), (2, 2688, 64, 20, 12),
                                   (9, 64, 128, 64, 64), (5, 128, 64, 126, 90)])   # the last two (and the second): quarter-split last round
def test_conv3x3_winograd_kernel_vs_direct_and_fp64(hip_lib, shape):
    """ud_conv3x3_wino_nhwc_f32 / ud_conv3x3_wino_wgrad_nhwc_f32 (forward, data gradient, weight gradient, BatchNorm partial sums) on every tile-block shape, ragged / odd maps and
    channel counts off the 64-wide blocks: against an fp64 convolution, next to the direct fp32 MFMA kernel's own error.
    Tolerance 2e-5 of the output's max (the direct kernels' bound); measured 1-3e-6 for both."""
    from unidistill_amd.ops import conv2d_f32 as c
    B, cin, cout, H, W = shape
    torch.manual_seed(sum(shape))
    x = _cl(torch.randn(B, cin, H, W, device="cuda"))
    w = torch.randn(cout, cin, 3, 3, device="cuda") * (9 * cin) ** -0.5
    b = torch.randn(cout, device="cuda")
    gy = _cl(torch.randn(B, cout, H, W, device="cuda"))
    ref = F.conv2d(x.double(), w.double(), b.double(), padding=1)
    gref = F.conv_transpose2d(gy.double(), w.double(), padding=1)
    old = (c.USE_WINOGRAD, c.WINO_MIN_FILL, c.USE_WINO4)
    try:
        c.USE_WINOGRAD, c.WINO_MIN_FILL, c.USE_WINO4 = True, 0.0, False       # F(2x2) here; F(4x4): the next test
        assert c.wino_pays(H, W, cin, cout)
        y, (part, slices, rows) = c._launch3(x, w, b, bn_stats=True)
        y2 = c._launch3(x, w, b, relu=True)
        gx = c._launch3(gy, w, transposed=True) if cout % 8 == 0 else None
        gw = c.weight_grad(x, gy, w, 3)
        c.USE_WINOGRAD = False
        yd = c._launch3(x, w, b) if cin % 32 == 0 else None
    finally:
        c.USE_WINOGRAD, c.WINO_MIN_FILL, c.USE_WINO4 = old
    tol = 2e-5 * float(ref.abs().max())
    err = float((y.double() - ref).abs().max())
    assert err <= tol, (err, tol)
    if yd is not None:
        assert err <= 4 * float((yd.double() - ref).abs().max()) + 1e-6 * float(ref.abs().max())
    assert torch.equal(y2, torch.relu(y))
    if gx is not None:
        assert float((gx.double() - gref).abs().max()) <= 2e-5 * float(gref.abs().max())
    gwref = torch.nn.grad.conv2d_weight(x.double(), w.shape, gy.double(), padding=1)
    assert float((gw.double() - gwref).abs().max()) <= 1e-4 * float(gwref.abs().max())      # the direct weight-gradient kernels' bound
    st = part[:slices * cout * 2].view(slices, cout, 2).double().sum(0)
    assert rows == B * H * W
    ys = y.double()
    np.testing.assert_allclose(st[:, 0].cpu().numpy(), ys.sum((0, 2, 3)).cpu().numpy(), rtol=1e-5, atol=1e-3)
    np.testing.assert_allclose(st[:, 1].cpu().numpy(), (ys * ys).sum((0, 2, 3)).cpu().numpy(), rtol=1e-5, atol=1e-3)


@pytest.mark.parametrize("sk", [0, 2], ids=["whole-units", "stream-k-tail"])
@pytest.mark.parametrize("shape", [(1, 64, 64, 16, 16), (2, 128, 128, 180, 180), (1, 256, 256, 90, 90), (3, 64, 192, 64, 176),
                                   (2, 128, 64, 32, 88), (5, 256, 256, 16, 44), (1, 512, 64, 8, 22), (2, 72, 100, 27, 35),
                                   (1, 64, 2688, 36, 28), (1, 64, 64, 1, 1), (1, 8, 4, 5, 3), (2, 2688, 64, 20, 12),
                                   (9, 64, 128, 64, 64), (5, 128, 64, 126, 90), (4, 512, 64, 180, 180), (1, 24, 40, 13, 19)])
def test_conv3x3_winograd_f4_kernel_vs_fp64(hip_lib, shape, sk):
    """ud_conv3x3_wino4_nhwc_f32 -- Winograd F(4x4, 3x3): forward (+ bias, BatchNorm partial sums, ReLU) and data gradient on the
    shapes of the F(2x2) test above + a 276-unit layer (stream-K over all units) against an fp64 convolution.  Tolerance
    1e-4 of the output's max (F(4x4)'s transforms round ~5x coarser than F(2x2)'s: measured 0.3-5e-5, the direct kernel 1-3e-6);
    both schedules -- whole units, and a stream-K tail with partial tiles + k_wino4_fixup -- must agree to that bound and the
    stream-K result must be reproducible bit for bit (fixed summation order of the pieces)."""
    from unidistill_amd import _lib
    from unidistill_amd.ops import conv2d_f32 as c
    B, cin, cout, H, W = shape
    torch.manual_seed(sum(shape))
    x = _cl(torch.randn(B, cin, H, W, device="cuda"))
    w = torch.randn(cout, cin, 3, 3, device="cuda") * (9 * cin) ** -0.5
    b = torch.randn(cout, device="cuda")
    gy = _cl(torch.randn(B, cout, H, W, device="cuda"))
    ref = F.conv2d(x.double(), w.double(), b.double(), padding=1)
    gref = F.conv_transpose2d(gy.double(), w.double(), padding=1)
    old = (c.USE_WINOGRAD, c.WINO4_MIN_FILL, c.USE_WINO4)
    lib = _lib.load()
    try:
        c.USE_WINOGRAD, c.WINO4_MIN_FILL, c.USE_WINO4 = True, 0.0, True
        lib.ud_conv3x3_wino4_stream_k(sk)
        assert c.wino4_pays(H, W, cin, cout)
        _lib.prof_enable(True)
        _lib.prof_read("conv2d.k_conv3x3_wino4_f32", reset=True)
        y, (part, slices, rows) = c._launch3(x, w, b, bn_stats=True)
        assert _lib.prof_read("conv2d.k_conv3x3_wino4_f32")[1] == 1
        _lib.prof_enable(False)
        y2 = c._launch3(x, w, b, relu=True)
        y3 = c._launch3(x, w, b)
        gx = c._launch3(gy, w, transposed=True) if cout % 8 == 0 else None
    finally:
        _lib.prof_enable(False)
        lib.ud_conv3x3_wino4_stream_k(-1)
        c.USE_WINOGRAD, c.WINO4_MIN_FILL, c.USE_WINO4 = old
    tol = 1e-4 * float(ref.abs().max())
    err = float((y.double() - ref).abs().max())
    assert err <= tol, (err, tol)
    assert torch.equal(y2, torch.relu(y)) and torch.equal(y3, y)
    if gx is not None:
        assert float((gx.double() - gref).abs().max()) <= 1e-4 * float(gref.abs().max())
    st = part[:slices * cout * 2].view(slices, cout, 2).double().sum(0)
    assert rows == B * H * W
    ys = y.double()
    np.testing.assert_allclose(st[:, 0].cpu().numpy(), ys.sum((0, 2, 3)).cpu().numpy(), rtol=1e-5, atol=1e-3)
    np.testing.assert_allclose(st[:, 1].cpu().numpy(), (ys * ys).sum((0, 2, 3)).cpu().numpy(), rtol=1e-5, atol=1e-3)


@pytest.mark.parametrize("shape", [(1, 64, 64, 16, 16), (2, 128, 128, 180, 180), (1, 256, 256, 90, 90), (3, 64, 192, 64, 176),
                                   (2, 128, 64, 32, 88), (5, 256, 256, 16, 44), (1, 512, 64, 8, 22), (1, 64, 2688, 36, 28),
                                   (1, 64, 64, 1, 1), (2, 2688, 64, 20, 12), (9, 64, 128, 64, 64), (5, 128, 64, 126, 90),
                                   (2, 32, 64, 13, 19), (3, 96, 128, 37, 5)])
def test_conv3x3_winograd_f4_weight_gradient_vs_fp64(hip_lib, shape):
    """ud_conv3x3_wino4_wgrad_nhwc_f32 -- the weight gradient through the F(4x4, 3x3) form: dW against an fp64 weight gradient on
    the F(2x2) test's shapes the kernel takes (Cin % 32 == 0, Cout % 64 == 0), odd / ragged maps (edge stages: masked patch
    pieces, zeroed dy pixels), a one-pixel map, slice counts from 1 to 128.  Tolerance 1e-4 of max |dW| (the direct weight-gradient
    kernels' bound; measured 2e-6 .. 2.5e-5); two runs bit-identical (ordered slice sum)."""
    from unidistill_amd import _lib
    from unidistill_amd.ops import conv2d_f32 as c
    B, cin, cout, H, W = shape
    torch.manual_seed(sum(shape))
    x = _cl(torch.randn(B, cin, H, W, device="cuda"))
    gy = _cl(torch.randn(B, cout, H, W, device="cuda"))
    w = torch.randn(cout, cin, 3, 3, device="cuda")
    old = (c.USE_WINO4_WGRAD, c.WINO4_WGRAD_ALL, c.USE_WINO4, c.USE_WINOGRAD)
    try:
        c.USE_WINO4_WGRAD, c.WINO4_WGRAD_ALL, c.USE_WINO4, c.USE_WINOGRAD = True, True, True, True
        assert c.wino4_wgrad_pays(B, H, W, cin, cout)
        _lib.prof_enable(True)
        _lib.prof_read("conv2d.k_wgrad_wino4_f32", reset=True)
        gw = c.weight_grad(x, gy, w, 3)
        assert _lib.prof_read("conv2d.k_wgrad_wino4_f32")[1] == 1
        _lib.prof_enable(False)
        gw2 = c.weight_grad(x, gy, w, 3)
    finally:
        _lib.prof_enable(False)
        c.USE_WINO4_WGRAD, c.WINO4_WGRAD_ALL, c.USE_WINO4, c.USE_WINOGRAD = old
    ref = torch.nn.grad.conv2d_weight(x.double(), w.shape, gy.double(), padding=1)
    assert gw.shape == ref.shape
    assert float((gw.double() - ref).abs().max()) <= 1e-4 * float(ref.abs().max())
    assert torch.equal(gw, gw2)


def test_winograd_filters_follow_parameter_updates_that_skip_the_version_counter(hip_lib):
    """torch's fused optimizers step parameters without moving ``_version`` (measured on torch._fused_adamw_): the transformed
    Winograd filters of a TRAINABLE weight must therefore never come from a version-keyed cache; a frozen weight's may, and
    must follow a regular in-place update."""
    from unidistill_amd.ops import conv2d_f32 as c
    torch.manual_seed(5)
    x = _cl(torch.randn(2, 64, 36, 28, device="cuda"))
    gy = _cl(torch.randn(2, 64, 36, 28, device="cuda"))
    for trainable in (True, False):
        w = (torch.randn(64, 64, 3, 3, device="cuda") * 0.05).requires_grad_(trainable)
        assert c.wino_pays(36, 28, 64, 64)
        y1, g1 = c._launch3(x, w), c._launch3(gy, w, transposed=True)
        v = w._version
        if trainable:
            w.data.mul_(2.0)                       # an update the version counter does not see (what a fused optimizer does)
            assert w._version == v
        else:
            with torch.no_grad():
                w.mul_(2.0)
        y2, g2 = c._launch3(x, w), c._launch3(gy, w, transposed=True)
        assert torch.allclose(y2, 2 * y1, rtol=1e-5, atol=1e-6) and torch.allclose(g2, 2 * g1, rtol=1e-5, atol=1e-6), trainable


def test_fp32_frozen_trunk_fuses_conv_bn_relu_and_matches_the_unfused_path(hip_lib):
    """Eval-mode (frozen teacher) BaseBEVBackbone in fp32: conv + folded BatchNorm + ReLU run as ONE kernel per link (scale / shift in
    the Winograd or direct epilogue) -- same output as the conv followed by the streaming BatchNorm kernel, and no BatchNorm launch."""
    from unidistill_amd import _lib
    from unidistill_amd.layers import dense
    from unidistill_amd.layers.bev import BaseBEVBackbone
    torch.manual_seed(1)
    m = BaseBEVBackbone([2, 2], [1, 2], [64, 128], [1, 2], [64, 64], 64).cuda()
    with torch.no_grad():
        for mod in m.modules():
            if isinstance(mod, torch.nn.BatchNorm2d):
                mod.running_mean.normal_(0, 0.2)
                mod.running_var.uniform_(0.5, 1.5)
                mod.weight.uniform_(0.5, 1.5)
                mod.bias.normal_(0, 0.2)
    m.eval().requires_grad_(False)
    x = _cl(torch.randn(2, 64, 44, 36, device="cuda"))
    _lib.prof_read("bn_act.k_fwd", reset=True)
    _lib.prof_enable(True)
    with torch.no_grad():
        y, _ = m(x)
    torch.cuda.synchronize()
    _lib.prof_enable(False)
    fused_bn_launches = _lib.prof_read("bn_act.k_fwd")[1]
    old = dense._can_fuse_inference
    dense._can_fuse_inference = lambda *a: False
    try:
        with torch.no_grad():
            yr, _ = m(x)
    finally:
        dense._can_fuse_inference = old
    assert (y - yr).abs().max() <= 2e-5 * yr.abs().max()
    assert fused_bn_launches <= 3, fused_bn_launches        # only the strided conv / deblock links keep a separate BatchNorm pass


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(270336, 256, 64), (270336, 64, 64), (16896, 1024, 256)])
def test_f32_1x1_weight_gradient_is_the_same_in_150_launches(hip_lib, shape):
    """A race detector for the hand-scheduled fragment reads of ud_conv1x1_wgrad_mapped_nhwc_f32 (plain staging, both tile widths):
    150 launches on the same operands give 150 bit-identical gradients.  (Round 5: the 64-wide variant copied a fragment register
    while its asm-issued LDS read was in flight -- about 1 launch in 40 differed in one wave's 16 x 64 block; the static twin of
    this test is tests/test_asm_inflight_cpu.py.)"""
    import ctypes as ct
    from unidistill_amd import _lib
    lib = _lib.load()
    P, K, N = shape
    torch.manual_seed(P + K + N)
    x = torch.randn(P, K, device="cuda")
    gy = torch.randn(P, N, device="cuda")
    nbytes = lib.ud_conv1x1_wgrad_f32_workspace_bytes(P, K, N)
    ws = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    outs = []
    for _ in range(150):
        dw = torch.empty(N, K, device="cuda")
        _lib.check(lib.ud_conv1x1_wgrad_mapped_nhwc_f32(_lib.ptr(x), _lib.ptr(gy), _lib.ptr(dw), P, K, N, None, None, _lib.ptr(ws),
                                                        nbytes, _lib.stream_of(x)), "ud_conv1x1_wgrad_mapped_nhwc_f32")
        outs.append(dw)
    torch.cuda.synchronize()
    different = sum(not torch.equal(outs[0], o) for o in outs[1:])
    assert different == 0, f"{different} of 149 launches differ from the first"
