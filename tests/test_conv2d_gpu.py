"""Hand-written MFMA convolution kernels (channels-last bf16) vs torch.nn.functional.conv2d in fp32."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _mk(B, Cin, H, W, Cout, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, Cin, H, W, generator=g).bfloat16()
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) / (3 * Cin ** 0.5)).bfloat16().float()
    b = torch.randn(Cout, generator=g)
    return x, w, b


# the last three shapes make the per-launch tile-height choice (csrc/conv2d.hip: ceil(WGs / 256) x (rows + 1)) pick 6-row
# tiles (180 x 180), 8-row tiles (256 full-width workgroups) and the 8-row 64-channel-wide tile; the small maps run the
# 4-row variants
@pytest.mark.parametrize("B,Cin,H,W,Cout", [(1, 64, 8, 16, 64), (2, 64, 19, 21, 128), (1, 128, 30, 37, 192),
                                            (2, 256, 9, 5, 64), (1, 128, 180, 180, 128), (16, 64, 8, 256, 128),
                                            (16, 64, 8, 256, 64)])
def test_conv3x3_forward_backward(hip_lib, B, Cin, H, W, Cout):
    from unidistill_amd.ops import conv2d as c2
    x, w, b = _mk(B, Cin, H, W, Cout, Cin + Cout + H)
    dev = torch.device("cuda:0")
    xr = x.float().to(dev).requires_grad_(True)
    wr = w.to(dev).requires_grad_(True)
    br = b.to(dev).requires_grad_(True)
    ref = F.conv2d(xr, wr, br, 1, 1)
    gy = torch.randn(ref.shape, generator=torch.Generator().manual_seed(3)).bfloat16().float().to(dev)
    ref.backward(gy)
    xd = x.to(dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    wd = w.to(dev).requires_grad_(True)
    bd = b.to(dev).requires_grad_(True)
    assert c2.supported(xd, wd)
    y = c2.conv3x3(xd, wd, bd)
    assert y.dtype == torch.bfloat16 and y.is_contiguous(memory_format=torch.channels_last)
    # bf16 operands are exact here; the only rounding is the bf16 store of y (2^-9 relative)
    tol = 6e-3 * float(ref.detach().abs().max())
    np.testing.assert_allclose(y.detach().float().cpu().numpy(), ref.detach().cpu().numpy(), rtol=0, atol=tol)
    y.backward(gy.to(torch.bfloat16))
    np.testing.assert_allclose(xd.grad.float().cpu().numpy(), xr.grad.cpu().numpy(), rtol=0,
                               atol=6e-3 * float(xr.grad.abs().max()))
    np.testing.assert_allclose(wd.grad.cpu().numpy(), wr.grad.cpu().numpy(), rtol=0,
                               atol=2e-2 * float(wr.grad.abs().max()))
    np.testing.assert_allclose(bd.grad.cpu().numpy(), br.grad.cpu().numpy(), rtol=1e-3,
                               atol=1e-3 * float(br.grad.abs().max()))


def test_conv3x3_fused_epilogue(hip_lib):
    from unidistill_amd.ops import conv2d as c2
    x, w, b = _mk(2, 64, 20, 23, 128, 11)
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(4)
    scale, shift = torch.rand(128, generator=g) + 0.5, torch.randn(128, generator=g)
    res = torch.randn(2, 128, 20, 23, generator=g).bfloat16()
    ref = F.relu((F.conv2d(x.float(), w, b, 1, 1) * scale[None, :, None, None] + shift[None, :, None, None])
                 + res.float())
    y = c2.conv3x3_inference(x.to(dev).contiguous(memory_format=torch.channels_last), w.to(dev), b.to(dev),
                             scale.to(dev), shift.to(dev), res.to(dev), relu=True)
    np.testing.assert_allclose(y.float().cpu().numpy(), ref.numpy(), rtol=0, atol=6e-3 * float(ref.abs().max()))


def test_trunk_on_mfma_kernel_matches_library_path(hip_lib):
    """BaseBEVBackbone under bf16 autocast: the hand-written conv/BN path is as close to the fp32
    result as the library's bf16 path is (train forward/backward and fused eval)."""
    from unidistill_amd.layers.bev import BaseBEVBackbone
    from unidistill_amd.layers import dense
    torch.manual_seed(0)
    net = BaseBEVBackbone([1, 1], [1, 2], [64, 128], [1, 2], [64, 64], 64).cuda()
    for m in net.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.normal_(0, 0.2); m.running_var.uniform_(0.5, 1.5)
    state = {k: v.clone() for k, v in net.state_dict().items()}
    x = torch.randn(2, 64, 36, 28, device="cuda").contiguous(memory_format=torch.channels_last)

    def run(hip, autocast):
        dense.Conv2d.hip_enabled = hip
        try:
            net.load_state_dict(state)
            net.train(); net.zero_grad()
            xs = x.clone().requires_grad_(True)
            with torch.autocast("cuda", dtype=torch.bfloat16, enabled=autocast):
                y, _ = net(xs)
            y.float().square().mean().backward()
            g = [p.grad.clone() for p in net.parameters()]
            net.eval()
            with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16, enabled=autocast):
                ye, _ = net(x)
            return [y.detach().float(), xs.grad.float(), torch.cat([t.flatten() for t in g]), ye.float()]
        finally:
            dense.Conv2d.hip_enabled = True

    ref, lib, ours = run(False, False), run(False, True), run(True, True)

    def rel(a, b):
        return float((a - b).norm()) / (float(b.norm()) + 1e-12)
    for name, r, l, o in zip(("train output", "input grad", "parameter grads", "eval output"), ref, lib, ours):
        e_lib, e_ours = rel(l, r), rel(o, r)
        assert e_ours < 1.5 * e_lib + 5e-3, (name, e_ours, e_lib)


def test_conv3x3_matches_numpy_oracle(hip_lib):
    """The CPU oracle's plain-numpy convolution (float64) vs the MFMA kernel on bf16-valued data."""
    import oracle
    from unidistill_amd.ops import conv2d as c2
    rng = np.random.default_rng(5)
    x = torch.from_numpy(rng.standard_normal((2, 11, 19, 128)).astype(np.float32)).cuda().bfloat16()
    w = torch.from_numpy((rng.standard_normal((64, 3, 3, 128)) * 0.04).astype(np.float32)).cuda().bfloat16()
    got = c2._launch(x.permute(0, 3, 1, 2), w.contiguous(), 64).permute(0, 2, 3, 1).float().cpu().numpy()
    ref = oracle.conv3x3_nhwc(x.float().cpu().numpy(), w.float().cpu().numpy())
    np.testing.assert_allclose(got, ref, rtol=0, atol=6e-3 * np.abs(ref).max())


@pytest.mark.parametrize("B,Cin,H,W,Cout", [(1, 64, 8, 16, 64), (2, 64, 64, 41, 256), (3, 256, 33, 17, 64),
                                            (2, 128, 20, 20, 368), (1, 512, 7, 9, 8), (2, 1024, 16, 44, 256),
                                            (1, 64, 37, 53, 64), (3, 128, 40, 41, 72), (1, 256, 64, 176, 64)])
def test_conv1x1_forward_backward(hip_lib, B, Cin, H, W, Cout):
    """1x1 convolution vs fp32 autograd: forward / data gradient on ud_conv1x1_nhwc_bf16 for maps above 1024
    pixels (pixel counts that are not multiples of the 128-pixel tile included) and on the library GEMM
    below, weight gradient on the pixel-reduced MFMA kernel ud_conv1x1_wgrad_nhwc_bf16 (bitwise repeatable)."""
    from unidistill_amd.ops import conv2d as c2
    g = torch.Generator().manual_seed(Cin + Cout + H)
    dev = torch.device("cuda:0")
    x = torch.randn(B, Cin, H, W, generator=g).bfloat16()
    w = (torch.randn(Cout, Cin, 1, 1, generator=g) / Cin ** 0.5).bfloat16().float()
    b = torch.randn(Cout, generator=g)
    xr, wr, br = x.float().to(dev).requires_grad_(True), w.to(dev).requires_grad_(True), b.to(dev).requires_grad_(True)
    ref = F.conv2d(xr, wr, br)
    gy = torch.randn(ref.shape, generator=g).bfloat16().float().to(dev)
    ref.backward(gy)
    xd = x.to(dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    wd, bd = w.to(dev).requires_grad_(True), b.to(dev).requires_grad_(True)
    assert c2.supported_1x1(xd, wd)
    y = c2.conv1x1(xd, wd, bd)
    assert y.dtype == torch.bfloat16
    np.testing.assert_allclose(y.detach().float().cpu().numpy(), ref.detach().cpu().numpy(), rtol=0,
                               atol=1e-2 * float(ref.detach().abs().max()))
    y.backward(gy.to(torch.bfloat16))
    np.testing.assert_allclose(xd.grad.float().cpu().numpy(), xr.grad.cpu().numpy(), rtol=0,
                               atol=1e-2 * float(xr.grad.abs().max()))
    # bf16 operands are exact, products are accumulated in fp32: only the summation order differs
    np.testing.assert_allclose(wd.grad.cpu().numpy(), wr.grad.cpu().numpy(), rtol=0,
                               atol=1e-4 * float(wr.grad.abs().max()))
    np.testing.assert_allclose(bd.grad.cpu().numpy(), br.grad.cpu().numpy(), rtol=1e-3,
                               atol=1e-3 * float(br.grad.abs().max()))
    xc, gc = xd.detach(), gy.to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    assert torch.equal(c2.weight_grad_1x1(xc, gc, wd.detach()), c2.weight_grad_1x1(xc, gc, wd.detach()))


def test_conv3x3_weight_gradient_large_map_kernel(hip_lib):
    """The tap-shared LDS-DMA weight-gradient kernel (maps with more than 4096 pixels) on ragged sizes:
    image edges that are not multiples of the 8 x 16 pixel tile, Cout that is not a multiple of 64."""
    from unidistill_amd.ops import conv2d as c2
    dev = torch.device("cuda:0")
    for (B, Cin, H, W, Cout) in [(1, 128, 67, 83, 72), (2, 64, 100, 90, 128), (5, 192, 33, 29, 136)]:
        g = torch.Generator().manual_seed(H + W)
        x = torch.randn(B, Cin, H, W, generator=g).bfloat16().to(dev).contiguous(memory_format=torch.channels_last)
        gy = torch.randn(B, Cout, H, W, generator=g).bfloat16().to(dev).contiguous(memory_format=torch.channels_last)
        w = torch.zeros(Cout, Cin, 3, 3, device=dev, requires_grad=True)
        F.conv2d(x.float(), w, None, 1, 1).backward(gy.float())
        c2_prev, c2.USE_HIP_WGRAD = c2.USE_HIP_WGRAD, True
        try:
            got = c2.weight_grad(x, gy, w.detach())
            again = c2.weight_grad(x, gy, w.detach())
        finally:
            c2.USE_HIP_WGRAD = c2_prev
        assert torch.equal(got, again)
        np.testing.assert_allclose(got.cpu().numpy(), w.grad.cpu().numpy(), rtol=0,
                                   atol=1e-4 * float(w.grad.abs().max()))


def test_conv1x1_fused_epilogue(hip_lib):
    from unidistill_amd.ops import conv2d as c2
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(8)
    B, Cin, H, W, Cout = 2, 192, 31, 47, 136
    x = torch.randn(B, Cin, H, W, generator=g).bfloat16()
    w = (torch.randn(Cout, Cin, generator=g) / Cin ** 0.5).bfloat16()
    b, scale, shift = torch.randn(Cout, generator=g), torch.rand(Cout, generator=g) + 0.5, torch.randn(Cout, generator=g)
    res = torch.randn(B, Cout, H, W, generator=g).bfloat16()
    ref = F.relu((F.conv2d(x.float(), w.float()[:, :, None, None], b) * scale[None, :, None, None]
                  + shift[None, :, None, None]) + res.float())
    cl = lambda t: t.to(dev).contiguous(memory_format=torch.channels_last)
    y = c2._launch1x1(cl(x), w.to(dev), Cout, b.to(dev), scale.to(dev), shift.to(dev), cl(res), relu=True)
    assert y.is_contiguous(memory_format=torch.channels_last)
    np.testing.assert_allclose(y.float().cpu().numpy(), ref.numpy(), rtol=0, atol=6e-3 * float(ref.abs().max()))


@pytest.mark.parametrize("H,W", [(12, 20), (40, 33)])
def test_conv_transpose_1x1_routes_to_conv1x1(hip_lib, H, W):
    """dense.ConvTranspose2d(k=1, s=1) -- the stride-1 deblocks -- equals nn.ConvTranspose2d (fp32) in
    forward, input gradient and weight gradient, on the GEMM (small map) and kernel (large map) paths."""
    from unidistill_amd.layers import dense
    dev = torch.device("cuda:0")
    torch.manual_seed(H)
    ours = dense.ConvTranspose2d(128, 72, 1, stride=1, bias=False).to(dev)
    ref = torch.nn.ConvTranspose2d(128, 72, 1, stride=1, bias=False).to(dev)
    with torch.no_grad():
        ours.weight.copy_(ours.weight.bfloat16().float())
        ref.weight.copy_(ours.weight)
    x = torch.randn(2, 128, H, W, device=dev).bfloat16()
    xr = x.float().requires_grad_(True)
    xo = x.contiguous(memory_format=torch.channels_last).requires_grad_(True)
    yr = ref(xr)
    gy = torch.randn_like(yr).bfloat16().float()
    yr.backward(gy)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        yo = ours(xo)
    assert yo.dtype == torch.bfloat16 and yo.shape == yr.shape
    yo.backward(gy.to(torch.bfloat16))
    tol = lambda t: 1e-2 * float(t.abs().max())
    np.testing.assert_allclose(yo.detach().float().cpu().numpy(), yr.detach().cpu().numpy(), rtol=0, atol=tol(yr))
    np.testing.assert_allclose(xo.grad.float().cpu().numpy(), xr.grad.cpu().numpy(), rtol=0, atol=tol(xr.grad))
    assert ours.weight.grad.shape == ref.weight.grad.shape
    np.testing.assert_allclose(ours.weight.grad.cpu().numpy(), ref.weight.grad.cpu().numpy(), rtol=0,
                               atol=1e-4 * float(ref.weight.grad.abs().max()))


@pytest.mark.parametrize("H,W", [(10, 12), (40, 41)])
def test_conv1x1_skip_adds_identity_gradient_in_kernel(hip_lib, H, W):
    """conv1x1_skip: the gradient of the identity branch is added in the data-gradient epilogue (large maps)
    or the GEMM's beta term (small maps): same result as conv + separate residual add."""
    from unidistill_amd.ops import conv2d as c2
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(W)
    x = torch.randn(2, 128, H, W, generator=g).bfloat16()
    w = (torch.randn(64, 128, 1, 1, generator=g) / 128 ** 0.5).bfloat16().float()
    w2 = (torch.randn(128, 64, 1, 1, generator=g) / 8).bfloat16().float()
    xr, wr, w2r = x.float().to(dev).requires_grad_(True), w.to(dev).requires_grad_(True), w2.to(dev)
    out_r = F.conv2d(F.conv2d(xr, wr), w2r) + xr                       # bottleneck-like: conv -> conv, + identity
    gy = torch.randn(out_r.shape, generator=g).bfloat16().float().to(dev)
    out_r.backward(gy)
    xd = x.to(dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    wd = w.to(dev).requires_grad_(True)
    y, idt = c2.conv1x1_skip(xd, wd)
    out = F.conv2d(y.float(), w2r) + idt.float()
    out.backward(gy)
    np.testing.assert_allclose(out.detach().cpu().numpy(), out_r.detach().cpu().numpy(), rtol=0,
                               atol=1.5e-2 * float(out_r.abs().max()))
    np.testing.assert_allclose(xd.grad.float().cpu().numpy(), xr.grad.cpu().numpy(), rtol=0,
                               atol=1.5e-2 * float(xr.grad.abs().max()))
    np.testing.assert_allclose(wd.grad.cpu().numpy(), wr.grad.cpu().numpy(), rtol=0,
                               atol=2e-2 * float(wr.grad.abs().max()))


def _cl(t):
    return t.contiguous(memory_format=torch.channels_last)


def _cmp(a, b, tol, name):
    a, b = a.float(), b.float()
    assert (a - b).abs().max() <= tol * b.abs().max() + 1e-6, (name, float((a - b).abs().max()), float(b.abs().max()))


@pytest.mark.parametrize("cfg", [(2, 256, 128, 4, 16, 44), (3, 512, 128, 2, 8, 22), (1, 64, 64, 2, 10, 18), (2, 128, 72, 2, 9, 7)])
def test_patch_conv_k_equals_stride(hip_lib, cfg, lenient):      # cfg3 (Cout = 72): the data gradient is a library op by design
    """nn.Conv2d(k = s, stride = s) on the mapped 1x1 kernels: forward, data and weight gradient vs PyTorch fp32 on
    the same bf16-valued tensors (odd sizes drop the remainder rows like the reference op)."""
    import torch.nn.functional as F
    from unidistill_amd.ops import conv2d as c
    B, cin, cout, s, H, W = cfg
    torch.manual_seed(sum(cfg))
    x = _cl(torch.randn(B, cin, H, W, device="cuda").bfloat16()).requires_grad_(True)
    w = (torch.randn(cout, cin, s, s, device="cuda") * 0.05).bfloat16().float().requires_grad_(True)
    y = c.conv_patch(x, w, s)
    gy = _cl(torch.randn_like(y))
    y.backward(gy)
    gx, gw = x.grad.clone(), w.grad.clone()
    x.grad = w.grad = None
    xr = x.detach().float().requires_grad_(True)
    yr = F.conv2d(xr, w, None, stride=s)
    yr.backward(gy.float())
    _cmp(y, yr, 8e-3, "y")
    _cmp(gx, xr.grad, 2e-2, "dx")
    _cmp(gw, w.grad, 2e-2, "dw")


@pytest.mark.parametrize("cfg", [(2, 256, 256, 2, 12, 20), (3, 2048, 128, 2, 8, 22), (1, 64, 64, 2, 5, 9)])
def test_patch_conv_transpose(hip_lib, cfg):
    import torch.nn.functional as F
    from unidistill_amd.ops import conv2d as c
    B, cin, cout, s, H, W = cfg
    torch.manual_seed(sum(cfg))
    x = _cl(torch.randn(B, cin, H, W, device="cuda").bfloat16()).requires_grad_(True)
    w = (torch.randn(cin, cout, s, s, device="cuda") * 0.05).bfloat16().float().requires_grad_(True)
    y = c.conv_transpose_patch(x, w, s)
    assert y.shape == (B, cout, H * s, W * s)
    gy = _cl(torch.randn_like(y))
    y.backward(gy)
    gx, gw = x.grad.clone(), w.grad.clone()
    x.grad = w.grad = None
    xr = x.detach().float().requires_grad_(True)
    yr = F.conv_transpose2d(xr, w, None, stride=s)
    yr.backward(gy.float())
    _cmp(y, yr, 8e-3, "y")
    _cmp(gx, xr.grad, 2e-2, "dx")
    _cmp(gw, w.grad, 2e-2, "dw")


@pytest.mark.parametrize("cfg", [(2, 256, 512, 2, 16, 44), (3, 1024, 2048, 2, 8, 22), (1, 64, 128, 2, 9, 7)])
def test_strided_1x1_conv(hip_lib, cfg):
    import torch.nn.functional as F
    from unidistill_amd.ops import conv2d as c
    B, cin, cout, s, H, W = cfg
    torch.manual_seed(sum(cfg))
    x = _cl(torch.randn(B, cin, H, W, device="cuda").bfloat16()).requires_grad_(True)
    w = (torch.randn(cout, cin, 1, 1, device="cuda") * 0.05).bfloat16().float().requires_grad_(True)
    y = c.conv1x1_strided(x, w, s)
    gy = _cl(torch.randn_like(y))
    y.backward(gy)
    gx, gw = x.grad.clone(), w.grad.clone()
    x.grad = w.grad = None
    xr = x.detach().float().requires_grad_(True)
    yr = F.conv2d(xr, w, None, stride=s)
    yr.backward(gy.float())
    _cmp(y, yr, 8e-3, "y")
    _cmp(gx, xr.grad, 2e-2, "dx")
    _cmp(gw, w.grad, 2e-2, "dw")


@pytest.mark.parametrize("cfg", [(2, 128, 128, 32, 88), (1, 128, 256, 45, 45), (2, 64, 64, 9, 7), (1, 256, 256, 16, 44), (1, 64, 128, 1, 5)])
def test_conv3x3_stride2(hip_lib, cfg):
    """3x3 / stride 2 / pad 1 on the mapped 1x1 kernels (im2col map, parity-class data gradient) vs PyTorch fp32."""
    import torch.nn.functional as F
    from unidistill_amd.ops import conv2d as c
    B, cin, cout, H, W = cfg
    torch.manual_seed(sum(cfg))
    x = _cl(torch.randn(B, cin, H, W, device="cuda").bfloat16()).requires_grad_(True)
    w = (torch.randn(cout, cin, 3, 3, device="cuda") * 0.05).bfloat16().float().requires_grad_(True)
    y = c.conv3x3_stride2(x, w)
    gy = _cl(torch.randn_like(y))
    y.backward(gy)
    gx, gw = x.grad.clone(), w.grad.clone()
    x.grad = w.grad = None
    xr = x.detach().float().requires_grad_(True)
    yr = F.conv2d(xr, w, None, stride=2, padding=1)
    yr.backward(gy.float())
    assert y.shape == yr.shape
    _cmp(y, yr, 8e-3, "y")
    _cmp(gx, xr.grad, 2e-2, "dx")
    _cmp(gw, w.grad, 2e-2, "dw")


@pytest.mark.gpu
def test_conv3x3_and_1x1_random_shapes_vs_fp32(hip_lib):
    """Seeded random shapes (ragged maps, channel counts off the tile widths, one to many slices) through the bf16 3x3 and
    1x1 kernels, forward and data gradient, against fp32 convolutions of the same bf16-valued tensors: exercises the edge
    tiles, the clamped rows / channels of the straight-line kernels and every tile-height variant the launcher can pick."""
    from unidistill_amd.ops import conv2d as c2
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(7)
    for it in range(24):
        B = int(rng.integers(1, 4))
        H, W = int(rng.integers(1, 70)), int(rng.integers(1, 70))
        Cin = 64 * int(rng.integers(1, 5))
        ks = 3 if it % 3 else 1
        # 3x3: the data gradient reduces over Cout in 64-channel slices (ops.conv2d.supported); 1x1: any multiple of 8
        Cout = 64 * int(rng.integers(1, 6)) if ks == 3 else 8 * int(rng.integers(1, 40))
        g = torch.Generator().manual_seed(100 + it)
        x = torch.randn(B, Cin, H, W, generator=g).bfloat16()
        w = (torch.randn(Cout, Cin, ks, ks, generator=g) * (Cin * ks * ks) ** -0.5).bfloat16().float()
        xr = x.float().to(dev).requires_grad_(True)
        wr = w.to(dev)
        ref = F.conv2d(xr, wr, None, 1, ks // 2)
        gy = torch.randn(ref.shape, generator=g).bfloat16().float().to(dev)
        ref.backward(gy)
        xd = x.to(dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
        wd = w.to(dev).requires_grad_(True)
        y = c2.conv3x3(xd, wd) if ks == 3 else c2.conv1x1(xd, wd)
        tag = f"case {it}: B={B} Cin={Cin} H={H} W={W} Cout={Cout} k={ks}"
        np.testing.assert_allclose(y.detach().float().cpu().numpy(), ref.detach().cpu().numpy(), rtol=0,
                                   atol=8e-3 * float(ref.detach().abs().max()) + 1e-6, err_msg=tag)
        y.backward(gy.to(torch.bfloat16))
        np.testing.assert_allclose(xd.grad.float().cpu().numpy(), xr.grad.cpu().numpy(), rtol=0,
                                   atol=8e-3 * float(xr.grad.abs().max()) + 1e-6, err_msg=tag)
