"""Detection-head tail (BN -> ReLU -> per-head conv) on the GPU vs a plain PyTorch fp32 composition
of the reference's modules (center_head.py:311-362: BatchNorm2d, ReLU, Conv2d per head)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _bf16(t):
    return t.bfloat16().float()


def _reference(y, gamma, beta, w2, b2, G, kmax, training, rm, rv, eps=1e-5, round_act=True):
    """fp32 torch: grouped conv over relu(bn(y)); activations/weights rounded to bf16 like the kernel."""
    a = F.relu(F.batch_norm(y, rm, rv, gamma, beta, training, 0.1, eps))
    if round_act:
        a = a + (_bf16(a) - a).detach()          # straight-through rounding
        w = w2 + (_bf16(w2) - w2).detach()
    else:
        w = w2
    return F.conv2d(a, w, b2, padding=1, groups=G)


def _case(B, H, W, G, kmax, seed):
    g = torch.Generator().manual_seed(seed)
    y = (torch.randn(B, G * 64, H, W, generator=g) * 1.5 + 0.3).bfloat16()
    gamma = torch.rand(G * 64, generator=g) + 0.5
    beta = torch.randn(G * 64, generator=g) * 0.2
    w2 = torch.randn(G * kmax, 64, 3, 3, generator=g) * 0.05
    if G * kmax > 1:
        w2[1] = 0                                     # a head narrower than kmax keeps zero rows
    b2 = torch.randn(G * kmax, generator=g)
    return y, gamma, beta, w2, b2


@pytest.mark.parametrize("B,H,W,G,kmax", [(2, 20, 36, 3, 3), (1, 16, 16, 2, 2), (3, 33, 17, 1, 1)])
@pytest.mark.parametrize("training", [True, False])
def test_head_tail_forward(hip_lib, B, H, W, G, kmax, training):
    from unidistill_amd.ops import head_tail as ht
    y, gamma, beta, w2, b2 = _case(B, H, W, G, kmax, 5)
    rm, rv = torch.randn(G * 64) * 0.1, torch.rand(G * 64) + 0.5
    rm_ref, rv_ref = rm.clone(), rv.clone()
    ref = _reference(y.float(), gamma, beta, w2, b2, G, kmax, training, rm_ref, rv_ref)
    dev = torch.device("cuda:0")
    rm_d, rv_d = rm.to(dev), rv.to(dev)
    yd = y.to(dev).contiguous(memory_format=torch.channels_last)
    z = ht.head_tail(yd, gamma.to(dev), beta.to(dev), w2.to(dev), b2.to(dev), rm_d, rv_d, training,
                     0.1, 1e-5, G, kmax)
    assert z.dtype == torch.float32 and z.shape == ref.shape
    # fp32 accumulation of bf16 products; a bf16 ulp flip of an activation moves z by ~1e-3
    np.testing.assert_allclose(z.cpu().numpy(), ref.numpy(), rtol=0, atol=4e-3 * float(ref.abs().max()))
    if training:   # nn.BatchNorm2d running-statistics bookkeeping
        np.testing.assert_allclose(rm_d.cpu().numpy(), rm_ref.numpy(), rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose(rv_d.cpu().numpy(), rv_ref.numpy(), rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("B,H,W,G,kmax", [(2, 20, 36, 3, 3), (1, 35, 18, 2, 2)])
def test_head_tail_backward(hip_lib, B, H, W, G, kmax):
    from unidistill_amd.ops import head_tail as ht
    y, gamma, beta, w2, b2 = _case(B, H, W, G, kmax, 9)
    leaves = [t.clone().requires_grad_(True) for t in (y.float(), gamma, beta, w2, b2)]
    ref = _reference(*leaves, G, kmax, True, None, None, round_act=False)
    gz = torch.randn(ref.shape, generator=torch.Generator().manual_seed(1))
    ref.backward(gz)
    dev = torch.device("cuda:0")
    yd = y.to(dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    prm = [t.to(dev).requires_grad_(True) for t in (gamma, beta, w2, b2)]
    z = ht.head_tail(yd, prm[0], prm[1], prm[2], prm[3], None, None, True, 0.1, 1e-5, G, kmax)
    z.backward(gz.to(dev))
    assert yd.grad.dtype == torch.bfloat16 and yd.grad.is_contiguous(memory_format=torch.channels_last)
    got = [yd.grad.float()] + [p.grad for p in prm]
    for name, a, r in zip(("dy", "dgamma", "dbeta", "dw2", "db2"), got, [l.grad for l in leaves]):
        tol = 2e-2 * float(r.abs().max())       # bf16 operands (dz, w2, activations) + bf16 dy
        np.testing.assert_allclose(a.cpu().numpy(), r.numpy(), rtol=0, atol=tol, err_msg=name)
    # deterministic: a second backward gives the same bits
    yd.grad = None
    for p in prm:
        p.grad = None
    z2 = ht.head_tail(yd, prm[0], prm[1], prm[2], prm[3], None, None, True, 0.1, 1e-5, G, kmax)
    z2.backward(gz.to(dev))
    assert torch.equal(z2, z) and torch.equal(yd.grad.float(), got[0]) and torch.equal(prm[2].grad, got[3])


def test_packed_heads_fused_tail_matches_library_path(hip_lib, lenient):
    """PackedSepHeads with the HIP tail == the same module on the MIOpen path (bf16 autocast)."""
    from unidistill_amd.layers.center_head import PackedSepHeads
    torch.manual_seed(0)
    heads = [{"reg": (2, 2), "height": (1, 2), "dim": (3, 2), "hm": (2, 2)}, {"reg": (2, 2), "hm": (1, 2)}]
    m = PackedSepHeads(64, heads).cuda()
    with torch.no_grad():
        m.c2_weight.mul_(0).add_(torch.randn_like(m.c2_weight) * 0.05 * (m.c2_weight != 0).any(dim=(1, 2, 3), keepdim=True))
    x = torch.randn(2, 64, 24, 20, device="cuda").contiguous(memory_format=torch.channels_last)
    outs = {}
    for fused in (True, False):
        m.fused_tail = fused
        m.zero_grad()
        xs = x.clone().requires_grad_(True)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            o = m(xs)
        loss = sum((v.float() ** 2).mean() for d in o for v in d.values())
        loss.backward()
        outs[fused] = ([v.float() for d in o for v in d.values()], xs.grad.float(), m.c1_weight.grad.clone(),
                       m.bn_weight.grad.clone(), m.c2_weight.grad.clone())
    for a, b in zip(outs[True][0], outs[False][0]):
        assert torch.allclose(a, b, rtol=0, atol=2e-2 * float(b.detach().abs().max()) + 1e-3)
    for a, b in zip(outs[True][1:], outs[False][1:]):
        assert torch.allclose(a, b, rtol=0, atol=4e-2 * float(b.detach().abs().max()) + 1e-6)


# the last shape is large enough for the 48-row strips of the real head (several 8-row blocks per workgroup, a partial last
# strip of 4 rows, a partial last column tile); the small ones run 8-row strips
@pytest.mark.parametrize("shape", [(2, 20, 36, 5, 3), (1, 17, 9, 42, 3), (2, 8, 16, 3, 4), (1, 33, 40, 2, 1),
                                   (2, 100, 140, 42, 3)])
def test_fp32_group_tail_vs_grouped_conv(hip_lib, shape):
    """fp32 mode: ud_head_tail_f32_fwd / _dgrad / _wgrad == torch's grouped 3x3 convolution (fp32), forward,
    input gradient, weight and bias gradients; 2e-5 of the max (summation order only)."""
    import torch
    import torch.nn.functional as F
    from unidistill_amd.ops import head_tail_f32 as h
    B, H, W, G, KM = shape
    torch.manual_seed(sum(shape))
    a = torch.randn(B, G * 64, H, W, device="cuda").contiguous(memory_format=torch.channels_last).requires_grad_(True)
    w = (torch.randn(G * KM, 64, 3, 3, device="cuda") * 0.1).requires_grad_(True)
    b = torch.randn(G * KM, device="cuda").requires_grad_(True)
    z = h.group_tail(a, w, b, G, KM)
    gz = torch.randn_like(z)
    z.backward(gz)
    got = (z.detach().clone(), a.grad.clone(), w.grad.clone(), b.grad.clone())
    a.grad = w.grad = b.grad = None
    zr = F.conv2d(a, w, b, padding=1, groups=G)
    zr.backward(gz)
    for x, y, name in zip(got, (zr.detach(), a.grad, w.grad, b.grad), ("z", "da", "dw", "db")):
        assert (x - y).abs().max() <= 2e-5 * y.abs().max() + 1e-6, name


@pytest.mark.parametrize("shape", [(2, 20, 36, 5, 3), (1, 17, 9, 42, 3), (2, 8, 16, 3, 4), (2, 100, 140, 42, 3)])
@pytest.mark.parametrize("training", [True, False])
def test_fp32_bn_relu_group_tail_vs_torch(hip_lib, shape, training):
    """ud_head_tail_f32_bn_fwd / _bn_wgrad (BatchNorm + ReLU applied as the tail kernels load the raw hidden tensor) + the
    BatchNorm backward fed by the tail's data gradient == F.batch_norm -> relu -> grouped conv3x3 in torch: output, running
    buffers, and the gradients of the hidden tensor, gamma, beta, tail weights and bias."""
    import torch
    import torch.nn.functional as F
    from unidistill_amd.ops import head_tail_f32 as h
    B, H, W, G, KM = shape
    C = G * 64
    torch.manual_seed(sum(shape))
    y = (torch.randn(B, C, H, W, device="cuda") * 1.5 + 0.3).contiguous(memory_format=torch.channels_last)
    gamma = (torch.rand(C, device="cuda") + 0.5).requires_grad_(True)
    beta = (torch.randn(C, device="cuda") * 0.2).requires_grad_(True)
    # the ReLU is not differentiable at 0 and the two sides round bn(y) differently: move the few elements that normalise to
    # within 1e-3 of 0 away from it (an element on the other branch shifts dy, dgamma and dbeta by its whole gradient)
    zn = F.batch_norm(y, None, None, gamma.detach(), beta.detach(), True, 0.0, 1e-5)
    y = (y + (zn.abs() < 1e-3).float() * 0.05 * torch.sign(gamma.detach()).view(1, -1, 1, 1)).requires_grad_(True)
    w = (torch.randn(G * KM, 64, 3, 3, device="cuda") * 0.1).requires_grad_(True)
    b = torch.randn(G * KM, device="cuda").requires_grad_(True)
    rm0, rv0 = torch.randn(C, device="cuda") * 0.1, torch.rand(C, device="cuda") + 0.5
    rm, rv = rm0.clone(), rv0.clone()
    z = h.bn_relu_group_tail(y, gamma, beta, rm, rv, training, 0.1, 1e-5, None, None, w, b, G, KM)
    rm_r, rv_r = rm0.clone(), rv0.clone()
    zr = F.conv2d(torch.relu(F.batch_norm(y, rm_r, rv_r, gamma, beta, training, 0.1, 1e-5)), w, b, padding=1, groups=G)
    assert (z - zr).abs().max() <= 2e-5 * zr.abs().max() + 1e-6
    assert torch.allclose(rm, rm_r, rtol=1e-5, atol=1e-6) and torch.allclose(rv, rv_r, rtol=1e-5, atol=1e-6)
    if not training:
        return
    gz = torch.randn_like(z)
    got = torch.autograd.grad(z, (y, gamma, beta, w, b), gz)
    ref = torch.autograd.grad(zr, (y, gamma, beta, w, b), gz)
    for a, r, name in zip(got, ref, ("dy", "dgamma", "dbeta", "dw", "db")):
        assert (a - r).abs().max() <= 1e-4 * r.abs().max() + 1e-6, name
    # the pass that stores dy also leaves its per-channel sums (the first convolution's bias gradient) on the tensor: equal to
    # a float64 sum of the stored values within fp32 summation error (the sums themselves are rounding noise around 0: the
    # gradient of a bias in front of a training-mode BatchNorm vanishes)
    from unidistill_amd.ops import bn_act
    dy = got[0]
    assert getattr(dy, "_ud_colsum", None) is not None
    sums = bn_act.bias_grad(dy)
    assert sums is dy._ud_colsum[2]
    ref64 = dy.double().sum((0, 2, 3))
    assert bool(((sums.double() - ref64).abs() <= 4e-6 * dy.double().abs().sum((0, 2, 3)) + 1e-6).all())


@pytest.mark.parametrize("fused_bn", [True, False])
def test_fp32_packed_heads_use_the_group_kernel_and_match_the_library_path(hip_lib, fused_bn):
    import torch
    from unidistill_amd import _lib
    from unidistill_amd.layers import center_head as ch
    from unidistill_amd.layers.dense import Conv2d
    torch.manual_seed(3)
    heads = {"reg": (2, 2), "height": (1, 2), "dim": (3, 2), "hm": (2, 2)}
    m = ch.PackedSepHeads(64, [heads, heads], head_conv=64, final_kernel=3).cuda().train()
    m.fused_bn_tail_f32 = fused_bn
    with torch.no_grad():
        m.c2_weight.normal_(0, 0.05)
    x = torch.randn(2, 64, 24, 20, device="cuda").contiguous(memory_format=torch.channels_last).requires_grad_(True)

    def run():
        m.zero_grad()
        x.grad = None
        m.bn_running_mean.zero_(); m.bn_running_var.fill_(1.0)
        outs = m(x)
        loss = sum((v ** 2).sum() for d in outs for v in d.values())
        loss.backward()
        return [v.detach().clone() for d in outs for v in d.values()], x.grad.clone(), m.c2_weight.grad.clone(), \
            m.c1_weight.grad.clone(), m.bn_weight.grad.clone()
    _lib.prof_read("head_tail.k_gtail_fwd", reset=True)
    _lib.prof_enable(True)
    ours = run()
    torch.cuda.synchronize()
    _lib.prof_enable(False)
    assert _lib.prof_read("head_tail.k_gtail_fwd")[1] == 1
    Conv2d.hip_enabled = False
    try:
        ref = run()
    finally:
        Conv2d.hip_enabled = True
    for a, b in zip(ours[0], ref[0]):
        assert (a - b).abs().max() <= 1e-4 * b.abs().max() + 1e-6
    for a, b, name in zip(ours[1:], ref[1:], ("dx", "dw2", "dw1", "dgamma")):
        assert (a - b).abs().max() <= 2e-4 * b.abs().max() + 1e-6, name
