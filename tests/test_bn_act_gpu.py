"""Streaming BatchNorm(+residual)(+ReLU) kernels vs torch.nn.BatchNorm2d in fp32."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _close(a, b, f, name=""):
    np.testing.assert_allclose(a.detach().float().cpu().numpy(), b.detach().float().cpu().numpy(), rtol=0,
                               atol=f * float(b.detach().abs().max()) + 1e-6, err_msg=name)


@pytest.mark.parametrize("B,C,H,W", [(2, 64, 9, 13), (3, 128, 16, 20), (1, 256, 5, 7)])
@pytest.mark.parametrize("residual", [False, True])
@pytest.mark.parametrize("relu", [True, False])
def test_bn_act_training(hip_lib, B, C, H, W, residual, relu):
    from unidistill_amd.layers.dense import batchnorm_act
    g = torch.Generator().manual_seed(C + H)
    x = (torch.randn(B, C, H, W, generator=g) * 2 + 0.5).bfloat16()
    r = torch.randn(B, C, H, W, generator=g).bfloat16() if residual else None
    gy = torch.randn(B, C, H, W, generator=g).bfloat16()
    bn_ref = torch.nn.BatchNorm2d(C, eps=1e-3, momentum=0.01)
    with torch.no_grad():
        bn_ref.weight.uniform_(0.5, 1.5); bn_ref.bias.normal_(0, 0.3)
    bn = torch.nn.BatchNorm2d(C, eps=1e-3, momentum=0.01).cuda()
    bn.load_state_dict(bn_ref.state_dict())
    xr = x.float().requires_grad_(True)
    rr = r.float().requires_grad_(True) if residual else None
    yr = bn_ref(xr)
    if residual:
        yr = yr + rr
    if relu:
        yr = F.relu(yr)
    yr.backward(gy.float())
    xd = x.cuda().contiguous(memory_format=torch.channels_last).requires_grad_(True)
    rd = r.cuda().contiguous(memory_format=torch.channels_last).requires_grad_(True) if residual else None
    y = batchnorm_act(bn, xd, rd, relu)
    assert y.dtype == torch.bfloat16 and y.is_contiguous(memory_format=torch.channels_last)
    y.backward(gy.cuda())
    _close(y, yr, 8e-3, "y")
    _close(xd.grad, xr.grad, 2e-2, "dx")
    _close(bn.weight.grad, bn_ref.weight.grad, 1e-2, "dgamma")
    _close(bn.bias.grad, bn_ref.bias.grad, 1e-2, "dbeta")
    if residual:
        _close(rd.grad, rr.grad, 8e-3, "dres")
    _close(bn.running_mean, bn_ref.running_mean, 1e-3, "running_mean")
    _close(bn.running_var, bn_ref.running_var, 1e-3, "running_var")
    assert int(bn.num_batches_tracked) == 1


def test_bn_act_eval(hip_lib):
    from unidistill_amd.layers.dense import batchnorm_act
    bn = torch.nn.BatchNorm2d(128).cuda().eval()
    with torch.no_grad():
        bn.running_mean.normal_(0, 0.5); bn.running_var.uniform_(0.5, 2); bn.weight.uniform_(0.5, 1.5)
        bn.bias.normal_(0, 0.2)
    x = torch.randn(2, 128, 11, 7, device="cuda").bfloat16().contiguous(memory_format=torch.channels_last)
    with torch.no_grad():
        y = batchnorm_act(bn, x)
        ref = F.relu(bn(x.float()))
    _close(y, ref, 8e-3)


@pytest.mark.parametrize("M,C", [(1000, 16), (777, 32), (300, 128), (5, 48)])
@pytest.mark.parametrize("residual", [False, True])
def test_bn_act_voxel_rows(hip_lib, M, C, residual):
    """BatchNorm1d (+ residual) + ReLU over sparse voxel rows [M, C] (spconv_backbone.py blocks)."""
    from unidistill_amd.layers.dense import batchnorm_act
    g = torch.Generator().manual_seed(M + C)
    x = (torch.randn(M, C, generator=g) * 1.5 - 0.2).bfloat16()
    r = torch.randn(M, C, generator=g).bfloat16() if residual else None
    gy = torch.randn(M, C, generator=g).bfloat16()
    bn_ref = torch.nn.BatchNorm1d(C, eps=1e-3, momentum=0.01)
    with torch.no_grad():
        bn_ref.weight.uniform_(0.5, 1.5); bn_ref.bias.normal_(0, 0.3)
    bn = torch.nn.BatchNorm1d(C, eps=1e-3, momentum=0.01).cuda()
    bn.load_state_dict(bn_ref.state_dict())
    xr = x.float().requires_grad_(True)
    rr = r.float().requires_grad_(True) if residual else None
    yr = F.relu(bn_ref(xr) + rr if residual else bn_ref(xr))
    yr.backward(gy.float())
    xd = x.cuda().requires_grad_(True)
    rd = r.cuda().requires_grad_(True) if residual else None
    y = batchnorm_act(bn, xd, rd, True)
    assert y.dtype == torch.bfloat16 and y.shape == (M, C)
    y.backward(gy.cuda())
    _close(y, yr, 8e-3, "y")
    _close(xd.grad, xr.grad, 2e-2, "dx")
    _close(bn.weight.grad, bn_ref.weight.grad, 1e-2, "dgamma")
    _close(bn.bias.grad, bn_ref.bias.grad, 1e-2, "dbeta")
    if residual:
        _close(rd.grad, rr.grad, 8e-3, "dres")
    _close(bn.running_mean, bn_ref.running_mean, 1e-3, "running_mean")
    _close(bn.running_var, bn_ref.running_var, 1e-3, "running_var")


@pytest.mark.parametrize("shape", [(2, 64, 9, 13), (3, 128, 16, 20), (777, 32), (300, 128)])
@pytest.mark.parametrize("residual", [False, True])
@pytest.mark.parametrize("relu", [True, False])
def test_bn_act_fp32_mode(hip_lib, shape, residual, relu):
    """fp32 twins (ud_bn_*_f32: the reference-arithmetic mode) vs torch BatchNorm in fp32: 2e-5 of the max."""
    from unidistill_amd import _lib
    from unidistill_amd.layers.dense import batchnorm_act
    g = torch.Generator().manual_seed(sum(shape))
    C = shape[1]
    cl = (lambda t: t.contiguous(memory_format=torch.channels_last)) if len(shape) == 4 else (lambda t: t.contiguous())
    x = torch.randn(*shape, generator=g) * 2 + 0.5
    r = torch.randn(*shape, generator=g) if residual else None
    gy = torch.randn(*shape, generator=g)
    BN = torch.nn.BatchNorm2d if len(shape) == 4 else torch.nn.BatchNorm1d
    bn_ref = BN(C, eps=1e-3, momentum=0.01)
    with torch.no_grad():
        bn_ref.weight.uniform_(0.5, 1.5); bn_ref.bias.normal_(0, 0.3)
    bn = BN(C, eps=1e-3, momentum=0.01).cuda()
    bn.load_state_dict(bn_ref.state_dict())
    xr = x.clone().requires_grad_(True)
    rr = r.clone().requires_grad_(True) if residual else None
    yr = bn_ref(xr)
    if residual:
        yr = yr + rr
    if relu:
        yr = F.relu(yr)
    yr.backward(gy)
    xd = cl(x.cuda()).requires_grad_(True)
    rd = cl(r.cuda()).requires_grad_(True) if residual else None
    _lib.prof_read("bn_act.k_fwd", reset=True)
    _lib.prof_enable(True)
    y = batchnorm_act(bn, xd, rd, relu)
    torch.cuda.synchronize()
    _lib.prof_enable(False)
    assert _lib.prof_read("bn_act.k_fwd")[1] == 1, "the fp32 tensor must take the HIP kernel, not the library"
    assert y.dtype == torch.float32
    y.backward(cl(gy.cuda()))
    _close(y, yr, 2e-5, "y")
    _close(xd.grad, xr.grad, 1e-4, "dx")
    _close(bn.weight.grad, bn_ref.weight.grad, 1e-4, "dgamma")
    _close(bn.bias.grad, bn_ref.bias.grad, 1e-4, "dbeta")
    if residual:
        _close(rd.grad, rr.grad, 2e-5, "dres")
    _close(bn.running_mean, bn_ref.running_mean, 1e-5, "running_mean")
    _close(bn.running_var, bn_ref.running_var, 1e-5, "running_var")


@pytest.mark.gpu
@pytest.mark.parametrize("dtype,ks,shape", [(torch.bfloat16, 3, (2, 64, 33, 41, 128)), (torch.bfloat16, 1, (3, 128, 40, 37, 80)),
                                            (torch.float32, 3, (2, 64, 33, 41, 128)), (torch.float32, 1, (3, 64, 40, 37, 96)),
                                            (torch.bfloat16, 3, (1, 128, 180, 180, 128))])
def test_conv_epilogue_statistics_equal_the_separate_pass(hip_lib, dtype, ks, shape):
    """A convolution launched with bn_stats hands the per-tile (sum, sum of squares) of its output to the BatchNorm behind
    it (ud_conv*_bnstats_nhwc_* + ud_bn_stats_from_partials): same normalised output, batch statistics, running statistics
    and gradients as the convolution followed by the stand-alone statistics pass (ud_bn_stats*)."""
    from unidistill_amd.ops import conv2d as c16, conv2d_f32 as c32, bn_act as hb
    B, cin, H, W, cout = shape
    torch.manual_seed(sum(shape) + ks)
    dev = torch.device("cuda:0")
    x0 = (torch.randn(B, cin, H, W, device=dev) + 0.3).to(dtype).contiguous(memory_format=torch.channels_last)
    w0 = torch.randn(cout, cin, ks, ks, device=dev) * (cin * ks * ks) ** -0.5
    gy = torch.randn(B, cout, H, W, device=dev).to(dtype).contiguous(memory_format=torch.channels_last)
    mod = c16 if dtype == torch.bfloat16 else c32
    conv = mod.conv3x3 if ks == 3 else mod.conv1x1
    outs = []
    for fused in (False, True):
        bn = torch.nn.BatchNorm2d(cout).to(dev).train()
        with torch.no_grad():
            bn.weight.copy_(torch.linspace(0.5, 1.5, cout)); bn.bias.copy_(torch.linspace(-0.5, 0.5, cout))
        x = x0.clone().requires_grad_(True)
        w = w0.clone().requires_grad_(True)
        y = conv(x, w, None, fused)
        assert hasattr(y, "_ud_bn_partial") == fused
        z = hb.bn_act(bn, y, None, True)
        z.backward(gy)
        outs.append([t.detach().float() for t in (z, bn.running_mean, bn.running_var, x.grad, w.grad, bn.weight.grad)])
    for a, b in zip(*outs):
        tol = (2e-2 if dtype == torch.bfloat16 else 2e-4) * float(a.abs().max()) + 1e-6
        np.testing.assert_allclose(b.cpu().numpy(), a.cpu().numpy(), rtol=0, atol=tol)
    # the statistics themselves agree far tighter than the bf16 tensors built from them
    np.testing.assert_allclose(outs[1][1].cpu().numpy(), outs[0][1].cpu().numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(outs[1][2].cpu().numpy(), outs[0][2].cpu().numpy(), rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("widths,shape", [((64, 32), (2, 9, 13)), ((256, 256), (2, 20, 36)), ((16, 48, 64), (3, 5, 7))])
def test_bn_act_into_a_concatenation_and_strided_gradient(hip_lib, dtype, widths, shape):
    """bn_act(..., out=slice) + cat_slices == torch.cat of separate bn_act calls, bit for bit (forward, dx, dgamma, dbeta, running
    statistics): ud_bn_act_fwd_ld writes rows of a wider map, ud_bn_act_bwd_ld reads the concatenation's gradient in place; and
    the plain torch.cat graph, whose backward hands over strided slices, takes the in-place path too (no copy)."""
    from unidistill_amd.ops import bn_act as hb
    B, H, W = shape
    g = torch.Generator().manual_seed(sum(widths) + H)
    cl = lambda t: t.contiguous(memory_format=torch.channels_last)
    xs = [cl((torch.randn(B, c, H, W, generator=g) * 2 + 0.3).cuda().to(dtype)) for c in widths]
    gy = cl(torch.randn(B, sum(widths), H, W, generator=g).cuda().to(dtype))

    def run(fused):
        bns = []
        for c in widths:
            bn = torch.nn.BatchNorm2d(c, eps=1e-3, momentum=0.01).cuda()
            with torch.no_grad():
                bn.weight.copy_(torch.linspace(0.5, 1.5, c)); bn.bias.copy_(torch.linspace(-0.3, 0.3, c))
            bns.append(bn)
        xin = [x.clone().requires_grad_(True) for x in xs]
        if fused:
            buf, slots = hb.cat_buffer(xin[0], list(widths))
            parts = [hb.bn_act(bn, x, None, True, out=s) for bn, x, s in zip(bns, xin, slots)]
            assert all(p.data_ptr() == s.data_ptr() for p, s in zip(parts, slots))
            y = hb.cat_slices(buf, parts)
        else:
            y = torch.cat([hb.bn_act(bn, x, None, True) for bn, x in zip(bns, xin)], 1)
        y.backward(gy)
        return [y.detach()] + [x.grad for x in xin] + [bn.weight.grad for bn in bns] + [bn.bias.grad for bn in bns] + \
            [bn.running_var for bn in bns]

    real = hb._like
    copies = []
    hb._like = lambda t, ref: (copies.append(tuple(t.shape)) if not t.is_contiguous(memory_format=torch.channels_last) else None,
                               real(t, ref))[1]
    try:
        a, b = run(True), run(False)
    finally:
        hb._like = real
    assert not copies, f"a strided gradient slice was copied: {copies}"
    assert a[0].is_contiguous(memory_format=torch.channels_last)
    for u, v in zip(a, b):
        assert torch.equal(u, v)


def test_bn_act_ld_argument_checks(hip_lib):
    from unidistill_amd import _lib
    lib = _lib.load()
    x = torch.randn(64, 32, device="cuda")
    y = torch.empty(64, 64, device="cuda")
    sc = torch.ones(32, device="cuda")
    s = _lib.stream_of(x)
    assert lib.ud_bn_act_fwd_ld_f32(x.data_ptr(), None, sc.data_ptr(), sc.data_ptr(), y.data_ptr(), 64, 32, 64, 1, s) == 0
    assert lib.ud_bn_act_fwd_ld_f32(x.data_ptr(), None, sc.data_ptr(), sc.data_ptr(), y.data_ptr(), 64, 32, 16, 1, s) != 0   # ld < C
    assert lib.ud_bn_act_fwd_ld_f32(x.data_ptr(), None, sc.data_ptr(), sc.data_ptr(), y.data_ptr(), 64, 32, 36, 1, s) != 0   # ld % 8
    assert lib.ud_bn_act_fwd_ld_f32(x.data_ptr(), None, sc.data_ptr(), sc.data_ptr(), y.data_ptr() + 4, 64, 32, 64, 1, s) != 0  # alignment
    torch.cuda.synchronize()
    assert torch.equal(y[:, :32], torch.relu(x + 1))


@pytest.mark.parametrize("autocast", [None, torch.bfloat16])
def test_bev_backbone_fused_concatenation_equals_torch_cat(hip_lib, autocast):
    """BaseBEVBackbone (base_bev_backbone.py:117-141) with the deblock outputs written into the concatenated map vs torch.cat."""
    from unidistill_amd.layers.bev import BaseBEVBackbone
    torch.manual_seed(3)
    net = BaseBEVBackbone([1, 1], [1, 2], [64, 128], [1, 2], [128, 128], 64).cuda().train()
    x = torch.randn(2, 64, 24, 40, device="cuda").contiguous(memory_format=torch.channels_last)
    outs = []
    for fuse in (True, False):
        BaseBEVBackbone.fuse_cat = fuse
        try:
            for p in net.parameters():
                p.grad = None
            xin = x.clone().requires_grad_(True)
            with torch.autocast("cuda", dtype=autocast, enabled=autocast is not None):
                y, _ = net(xin)
            y.float().square().mean().backward()
            outs.append([y.detach().float(), xin.grad] + [p.grad.clone() for p in net.parameters()])
        finally:
            BaseBEVBackbone.fuse_cat = True
    assert outs[0][0].shape == (2, 256, 24, 40)
    for u, v in zip(*outs):
        assert torch.equal(u, v)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", ["f32", "bf16"])
@pytest.mark.parametrize("shape", [(4, 64, 45, 45), (2, 126, 60, 44), (1, 2688, 30, 20), (2, 16, 33, 17), (3, 368, 16, 44),
                                   (1, 8, 1, 1), (2, 300, 9, 7)])
def test_bias_gradient_column_sums(hip_lib, dtype, shape):
    """ud_colsum_f32 / _bf16 (the bias gradient of a convolution: gy.sum((0, 2, 3)), center_head.py:64,339,353) vs a float64 sum:
    within fp32 summation error of the column's absolute sum, bitwise the same on a second call, also for [M, C] row tensors and
    channel counts that take the scalar kernel (126 = the 42 packed heads x 3)."""
    from unidistill_amd.ops import bn_act
    B, C, H, W = shape
    dt = torch.float32 if dtype == "f32" else torch.bfloat16
    torch.manual_seed(C + H)
    g = (torch.randn(B, C, H, W, device="cuda") * 2 + 0.1).to(dt).contiguous(memory_format=torch.channels_last)
    got = bn_act.bias_grad(g)
    ref = g.double().sum((0, 2, 3))
    bound = 4e-6 * g.double().abs().sum((0, 2, 3)) + 1e-6
    assert got.dtype == torch.float32 and got.shape == (C,)
    assert bool(((got.double() - ref).abs() <= bound).all())
    assert torch.equal(got, bn_act.bias_grad(g))
    rows = g.permute(0, 2, 3, 1).reshape(-1, C)
    assert rows.is_contiguous()
    assert torch.equal(bn_act.bias_grad(rows), got)


@pytest.mark.gpu
def test_bias_gradient_sums_left_by_the_producer_are_used_and_invalidated(hip_lib):
    from unidistill_amd.ops import bn_act
    g = torch.randn(2, 32, 8, 8, device="cuda").contiguous(memory_format=torch.channels_last)
    marker = torch.full((32,), 7.0, device="cuda")
    bn_act.attach_colsum(g, marker)
    assert bn_act.bias_grad(g) is marker
    g.add_(1.0)                                            # version moved: the recorded sums are stale
    assert torch.allclose(bn_act.bias_grad(g), g.sum((0, 2, 3)), rtol=1e-5, atol=1e-4)
    bn_act.attach_colsum(g, marker)
    bn_act.drop_colsum(g)                                  # (what a raw-pointer in-place writer calls)
    assert bn_act.bias_grad(g) is not marker
