"""GPU parity: sparse conv index/rulebooks (bit-exact) and conv/dgrad/wgrad (fp32 tolerance)
vs the CPU oracle and vs dense torch conv3d."""
import numpy as np
import pytest
import torch

import oracle

pytestmark = pytest.mark.gpu


def _sites(rng, B, Dz, Hy, Wx, p, shuffle=True):
    occ = rng.random((B, Dz, Hy, Wx)) < p
    coords = np.argwhere(occ).astype(np.int32)
    if shuffle:
        rng.shuffle(coords)
    return coords


def _tensor(coords, feat, shape):
    from unidistill_amd.ops import spconv as sp
    return sp.SparseConvTensor(torch.from_numpy(feat).cuda(), torch.from_numpy(coords).cuda(),
                               shape[1:], shape[0])


@pytest.mark.parametrize("shape,p,ks", [((2, 6, 9, 8), 0.3, (3, 3, 3)),
                                         ((1, 41, 60, 70), 0.02, (3, 3, 3)),
                                         ((3, 5, 17, 130), 0.5, (3, 1, 1)),
                                         ((1, 2, 3, 3), 1.0, (3, 3, 3))])
def test_subm_rulebook_bitexact(shape, p, ks):
    rng = np.random.default_rng(sum(shape))
    coords = _sites(rng, *shape, p)
    x = _tensor(coords, np.zeros((len(coords), 4), np.float32), shape)
    nbr = x._sites.subm_rulebook(ks).cpu().numpy()
    np.testing.assert_array_equal(nbr, oracle.spconv_subm_rulebook(coords, shape, ks))


@pytest.mark.parametrize("shape,p,ks,st,pd", [
    ((2, 6, 9, 8), 0.3, (3, 3, 3), (2, 2, 2), (1, 1, 1)),
    ((1, 41, 64, 64), 0.01, (3, 3, 3), (2, 2, 2), (1, 1, 1)),
    ((2, 11, 20, 20), 0.1, (3, 3, 3), (2, 2, 2), (0, 1, 1)),     # conv4 padding (0,1,1)
    ((2, 5, 12, 12), 0.2, (3, 1, 1), (2, 1, 1), (0, 0, 0)),      # conv_out
])
def test_down_rulebooks_bitexact(shape, p, ks, st, pd):
    rng = np.random.default_rng(sum(shape) + 1)
    coords = _sites(rng, *shape, p)
    x = _tensor(coords, np.zeros((len(coords), 4), np.float32), shape)
    out_sites, out_nbr, in_nbr = x._sites.down(ks, st, pd)
    oc, onbr, inbr, oshape = oracle.spconv_down(coords, shape, ks, st, pd)
    assert tuple(out_sites.spatial_shape) == tuple(oshape[1:])
    np.testing.assert_array_equal(out_sites.indices.cpu().numpy(), oc)     # sorted (b,z,y,x)
    np.testing.assert_array_equal(out_nbr.cpu().numpy(), onbr)
    np.testing.assert_array_equal(in_nbr.cpu().numpy(), inbr)
    # the output level's own index must be usable: a subm rulebook on it matches the oracle
    nb2 = out_sites.subm_rulebook((3, 3, 3)).cpu().numpy()
    np.testing.assert_array_equal(nb2, oracle.spconv_subm_rulebook(oc, oshape, (3, 3, 3)))


def _tol(ref):
    return dict(rtol=2e-5, atol=2e-5 * max(1.0, float(np.abs(ref).max())))


@pytest.mark.parametrize("cin,cout", [(5, 16), (16, 16), (16, 32), (32, 32), (32, 64), (64, 64),
                                      (64, 128), (128, 128), (7, 9)])
@pytest.mark.parametrize("algo", [0, 1, 2])
def test_conv_dgrad_wgrad_vs_oracle(cin, cout, algo):
    from unidistill_amd.ops import spconv as sp
    rng = np.random.default_rng(cin * 1000 + cout)
    shape = (2, 7, 18, 20)
    coords = _sites(rng, *shape, 0.25)
    M = len(coords)
    feat = rng.standard_normal((M, cin)).astype(np.float32)
    W = (rng.standard_normal((cout, 27, cin)) / np.sqrt(27 * cin)).astype(np.float32)
    bias = rng.standard_normal(cout).astype(np.float32)
    nbr = oracle.spconv_subm_rulebook(coords, shape, (3, 3, 3))
    conv = sp.SubMConv3d(cin, cout, 3, padding=1, bias=True).cuda()
    conv.kernel_algo = algo
    with torch.no_grad():
        conv.weight.copy_(torch.from_numpy(W).view(cout, 3, 3, 3, cin))
        conv.bias.copy_(torch.from_numpy(bias))
    x = _tensor(coords, feat, shape)
    x.features.requires_grad_(True)
    y = conv(x)
    ref = oracle.spconv_conv(feat, nbr, W, bias)
    np.testing.assert_allclose(y.features.detach().cpu().numpy(), ref, **_tol(ref))
    gout = rng.standard_normal(ref.shape).astype(np.float32)
    y.features.backward(torch.from_numpy(gout).cuda())
    ref_gin = oracle.spconv_conv(gout, nbr, W, mirror=True, transpose=True)
    np.testing.assert_allclose(x.features.grad.cpu().numpy(), ref_gin, **_tol(ref_gin))
    ref_gw = oracle.spconv_wgrad(feat, nbr, gout, cout)
    np.testing.assert_allclose(conv.weight.grad.cpu().numpy().reshape(cout, 27, cin), ref_gw, **_tol(ref_gw))
    np.testing.assert_allclose(conv.bias.grad.cpu().numpy(), gout.sum(0), rtol=1e-4, atol=1e-4)


def _bf16(a):
    """Round an fp32 array to bf16 (nearest even) and back -- what algo 3 does when it stages tiles."""
    u = np.ascontiguousarray(a, dtype=np.float32).view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    return u.astype(np.uint32).view(np.float32).reshape(a.shape)


@pytest.mark.parametrize("cin,cout", [(5, 16), (16, 32), (32, 32), (64, 64), (64, 128), (128, 128), (7, 9)])
def test_bf16_mfma_conv_equals_oracle_on_bf16_rounded_operands(cin, cout):
    """algo 3 rounds activations and weights to bf16 and accumulates in fp32: it must match the
    oracle fed the same rounded operands to fp32-accumulation accuracy (a far tighter check than
    a bf16-sized tolerance against the unrounded result).  Forward, data gradient and weight gradient."""
    from unidistill_amd.ops import spconv as sp
    rng = np.random.default_rng(cin * 77 + cout)
    shape = (2, 7, 18, 20)
    coords = _sites(rng, *shape, 0.3)
    M = len(coords)
    feat = rng.standard_normal((M, cin)).astype(np.float32)
    W = (rng.standard_normal((cout, 27, cin)) / np.sqrt(27 * cin)).astype(np.float32)
    bias = rng.standard_normal(cout).astype(np.float32)
    nbr = oracle.spconv_subm_rulebook(coords, shape, (3, 3, 3))
    conv = sp.SubMConv3d(cin, cout, 3, padding=1, bias=True).cuda()
    with torch.no_grad():
        conv.weight.copy_(torch.from_numpy(W).view(cout, 3, 3, 3, cin))
        conv.bias.copy_(torch.from_numpy(bias))
    x = _tensor(coords, feat, shape)
    x.features.requires_grad_(True)
    with torch.autocast("cuda", dtype=torch.bfloat16):     # default algo 0 -> 3 under autocast
        assert sp.effective_algo(0) == 3
        y = conv(x)
    bf_io = cin % 8 == 0 and cout % 8 == 0          # activations stay bf16 in HBM on this path
    assert y.features.dtype == (torch.bfloat16 if bf_io else torch.float32)
    def tol(r):                                      # + one bf16 rounding of the stored result
        return dict(rtol=0, atol=6e-3 * max(1.0, float(np.abs(r).max()))) if bf_io else _tol(r)
    ref = oracle.spconv_conv(_bf16(feat), nbr, _bf16(W), bias)
    np.testing.assert_allclose(y.features.detach().float().cpu().numpy(), ref, **tol(ref))
    full = oracle.spconv_conv(feat, nbr, W, bias)
    assert np.abs(y.features.detach().float().cpu().numpy() - full).max() < 3e-2 * max(1.0, np.abs(full).max())
    gout = rng.standard_normal(ref.shape).astype(np.float32)
    y.features.backward(torch.from_numpy(gout).cuda())
    ref_gin = oracle.spconv_conv(_bf16(gout), nbr, _bf16(W), mirror=True, transpose=True)
    np.testing.assert_allclose(x.features.grad.float().cpu().numpy(), ref_gin, **tol(ref_gin))
    ref_gw = oracle.spconv_wgrad(_bf16(feat), nbr, _bf16(gout), cout)
    np.testing.assert_allclose(conv.weight.grad.cpu().numpy().reshape(cout, 27, cin), ref_gw, **_tol(ref_gw))


def test_strided_conv_and_dense_vs_torch_conv3d():
    """SparseConv3d + .dense() == dense conv3d evaluated on the reachable set; grads too."""
    from unidistill_amd.ops import spconv as sp
    rng = np.random.default_rng(5)
    shape = (2, 9, 14, 16)
    coords = _sites(rng, *shape, 0.15)
    cin, cout = 16, 32
    feat = rng.standard_normal((len(coords), cin)).astype(np.float32)
    conv = sp.SparseConv3d(cin, cout, 3, stride=2, padding=1, bias=False, indice_key="d").cuda()
    x = _tensor(coords, feat, shape)
    x.features.requires_grad_(True)
    y = conv(x)
    dense = y.dense()
    # dense reference
    xd = torch.from_numpy(oracle.sparse_to_dense(feat, coords, shape)).cuda().requires_grad_(True)
    wt = conv.weight.detach().permute(0, 4, 1, 2, 3).contiguous().requires_grad_(True)
    ref = torch.nn.functional.conv3d(xd, wt, stride=2, padding=1)
    occ = torch.zeros((shape[0], 1) + shape[1:], device="cuda")
    occ[coords[:, 0], 0, coords[:, 1], coords[:, 2], coords[:, 3]] = 1
    reach = torch.nn.functional.conv3d(occ, torch.ones(1, 1, 3, 3, 3, device="cuda"), stride=2, padding=1) > 0
    ref_m = ref * reach
    np.testing.assert_allclose(dense.detach().cpu().numpy(), ref_m.detach().cpu().numpy(), rtol=1e-4, atol=1e-4)
    g = torch.randn_like(dense)
    dense.backward(g)
    ref_m.backward(g)
    gx_ref = xd.grad[coords[:, 0], :, coords[:, 1], coords[:, 2], coords[:, 3]]
    np.testing.assert_allclose(x.features.grad.cpu().numpy(), gx_ref.cpu().numpy(), rtol=1e-4, atol=1e-4)
    gw_ref = wt.grad.permute(0, 2, 3, 4, 1)
    np.testing.assert_allclose(conv.weight.grad.cpu().numpy(), gw_ref.cpu().numpy(), rtol=1e-4, atol=2e-4)


def test_inverse_conv_shapes_and_empty_input():
    from unidistill_amd.ops import spconv as sp
    rng = np.random.default_rng(6)
    shape = (1, 8, 12, 12)
    coords = _sites(rng, *shape, 0.2)
    x = _tensor(coords, rng.standard_normal((len(coords), 16)).astype(np.float32), shape)
    down = sp.SparseConv3d(16, 32, 3, stride=2, padding=1, indice_key="k").cuda()
    up = sp.SparseInverseConv3d(32, 16, 3, indice_key="k").cuda()
    z = up(down(x))
    assert z.features.shape == (len(coords), 16) and z.indices is x.indices
    empty = _tensor(np.zeros((0, 4), np.int32), np.zeros((0, 16), np.float32), shape)
    e = sp.SubMConv3d(16, 16, 3, padding=1).cuda()(empty)
    assert e.features.shape == (0, 16)
    assert e.dense().abs().sum().item() == 0


@pytest.mark.parametrize("shape,p,pd", [((2, 8, 12, 12), 0.2, (1, 1, 1)), ((1, 11, 20, 18), 0.1, (0, 1, 1))])
def test_inverse_conv_values_vs_oracle_and_dense_transposed_conv(shape, p, pd):
    """SparseInverseConv3d (spconv: the transposed rulebook of the SparseConv3d sharing its indice_key, SAME kernel
    offsets, outputs at the fine sites): values vs the oracle's conv over ``in_nbr`` and vs torch's dense
    conv_transpose3d sampled at the fine sites; gradients vs autograd through that dense formulation."""
    from unidistill_amd.ops import spconv as sp
    rng = np.random.default_rng(61 + shape[1])
    coords = _sites(rng, *shape, p)
    feat = rng.standard_normal((len(coords), 16)).astype(np.float32)
    x = _tensor(coords, feat, shape)
    x.features.requires_grad_(True)
    torch.manual_seed(2)
    down = sp.SparseConv3d(16, 32, 3, stride=2, padding=pd, indice_key="k").cuda()
    up = sp.SparseInverseConv3d(32, 16, 3, indice_key="k").cuda()
    mid = down(x)
    z = up(mid)
    assert z.features.shape == (len(coords), 16) and z.indices is x.indices
    oc, onbr, inbr, oshape = oracle.spconv_down(coords, shape, (3, 3, 3), (2, 2, 2), pd)
    Wd = down.weight.detach().cpu().numpy().reshape(32, 27, 16)
    Wu = up.weight.detach().cpu().numpy().reshape(16, 27, 32)
    d_ref = oracle.spconv_conv(feat, onbr, Wd, down.bias.detach().cpu().numpy())
    z_ref = oracle.spconv_conv(d_ref, inbr, Wu, up.bias.detach().cpu().numpy())
    np.testing.assert_allclose(z.features.detach().cpu().numpy(), z_ref, **_tol(z_ref))
    # independent formulation: dense coarse map -> conv_transpose3d -> sampled at the fine sites
    B, Dz, Hy, Wx = shape
    dm = torch.zeros((B, 32) + tuple(oshape[1:]), dtype=torch.float64, device="cuda")
    ocl = torch.from_numpy(oc).long().cuda()
    mid_leaf = mid.features.detach().double().requires_grad_(True)
    dm[ocl[:, 0], :, ocl[:, 1], ocl[:, 2], ocl[:, 3]] = mid_leaf
    wt = up.weight.detach().double().permute(4, 0, 1, 2, 3).contiguous().requires_grad_(True)   # [Cin, Cout, kz, ky, kx]
    opad = [d - ((o - 1) * 2 - 2 * q + 3) for d, o, q in zip((Dz, Hy, Wx), oshape[1:], pd)]
    dense = torch.nn.functional.conv_transpose3d(dm, wt, up.bias.detach().double(), stride=2, padding=pd,
                                                 output_padding=opad)
    cl = torch.from_numpy(coords).long().cuda()
    z_dense = dense[cl[:, 0], :, cl[:, 1], cl[:, 2], cl[:, 3]]
    np.testing.assert_allclose(z.features.detach().cpu().numpy(), z_dense.detach().float().cpu().numpy(), **_tol(z_ref))
    g = torch.from_numpy(rng.standard_normal(z_ref.shape).astype(np.float32)).cuda()
    mid.features.retain_grad()
    z.features.backward(g)
    z_dense.backward(g.double())
    gw_ref = wt.grad.permute(1, 2, 3, 4, 0).float().cpu().numpy()
    np.testing.assert_allclose(up.weight.grad.cpu().numpy(), gw_ref, rtol=1e-4, atol=2e-4 * max(1.0, np.abs(gw_ref).max()))
    gm_ref = mid_leaf.grad.float().cpu().numpy()
    np.testing.assert_allclose(mid.features.grad.cpu().numpy(), gm_ref, **_tol(gm_ref))
    assert x.features.grad is not None and torch.isfinite(x.features.grad).all()


def test_rulebook_shared_between_indice_keys():
    from unidistill_amd.ops import spconv as sp
    rng = np.random.default_rng(8)
    shape = (1, 6, 10, 10)
    coords = _sites(rng, *shape, 0.3)
    x = _tensor(coords, rng.standard_normal((len(coords), 16)).astype(np.float32), shape)
    a = sp.SubMConv3d(16, 16, 3, padding=1, indice_key="subm1").cuda()
    b = sp.SubMConv3d(16, 16, 3, padding=1, indice_key="res1").cuda()
    y = b(a(x))
    assert len(y._sites._subm) == 1          # one rulebook for both keys


def test_fused_inference_epilogue_matches_module_chain():
    """conv + eval BatchNorm1d + residual + ReLU fused in the kernel epilogue == the op chain."""
    from unidistill_amd.ops import spconv as sp
    from unidistill_amd.layers.lidar import SparseBasicBlock
    from functools import partial
    rng = np.random.default_rng(21)
    shape = (2, 7, 18, 20)
    coords = _sites(rng, *shape, 0.3)
    torch.manual_seed(3)
    norm = partial(torch.nn.BatchNorm1d, eps=1e-3, momentum=0.01)
    for c in (16, 64):
        feat = rng.standard_normal((len(coords), c)).astype(np.float32)
        blk = SparseBasicBlock(c, c, norm_fn=norm, indice_key="r").cuda().eval()
        seq = sp.SparseSequential(sp.SparseConv3d(c, 2 * c, 3, stride=2, padding=1, bias=False), norm(2 * c),
                                  torch.nn.ReLU()).cuda().eval()
        for m in list(blk.modules()) + list(seq.modules()):
            if isinstance(m, torch.nn.BatchNorm1d):
                m.running_mean.normal_(0, 0.3); m.running_var.uniform_(0.5, 2.0)
                m.weight.data.uniform_(0.5, 1.5); m.bias.data.normal_(0, 0.3)
        x = _tensor(coords, feat, shape)
        with torch.no_grad():
            fused = seq(blk(x)).features
        # unfused reference: force the autograd path by requiring grad on the input
        xr = _tensor(coords, feat, shape)
        xr.features.requires_grad_(True)
        ref = seq(blk(xr)).features
        assert ref.requires_grad and not fused.requires_grad
        np.testing.assert_allclose(fused.cpu().numpy(), ref.detach().cpu().numpy(), rtol=2e-5, atol=2e-5)
        # mixed precision: under bf16 autocast the fused chain keeps bf16 activations in HBM
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
            mixed = seq(blk(_tensor(coords, feat, shape)))
        assert mixed.features.dtype == torch.bfloat16
        err = (mixed.features.float() - ref.detach()).abs().max().item()
        assert err < 3e-2 * max(1.0, ref.detach().abs().max().item()), err
        assert mixed.dense().dtype == torch.float32


def test_bev_nhwc_bf16_equals_dense_view():
    """SparseConvTensor.bev() on bf16 features == dense().view(N, C*D, H, W) (pure data movement), fwd and bwd."""
    rng = np.random.default_rng(8)
    shape = (2, 2, 18, 20)
    coords = _sites(rng, *shape, 0.3)
    feat = rng.standard_normal((len(coords), 24)).astype(np.float32)
    x = _tensor(coords, feat, shape)
    fb = x.features.to(torch.bfloat16).detach().requires_grad_(True)
    xb = x.replace_feature(fb)
    bev = xb.bev()
    assert bev.dtype == torch.bfloat16 and bev.shape == (2, 48, 18, 20)
    assert bev.is_contiguous(memory_format=torch.channels_last)
    ff = fb.detach().float().requires_grad_(True)
    ref = x.replace_feature(ff).dense()
    ref = ref.view(2, 48, 18, 20)
    assert torch.equal(bev.float(), ref)
    g = torch.randn(2, 48, 18, 20, device="cuda").bfloat16()
    bev.backward(g)
    ref.backward(g.float())
    assert torch.equal(fb.grad.float(), ff.grad)


def test_bev_nhwc_f32_equals_dense_view():
    """fp32 features: SparseConvTensor.bev() (ud_sparse_to_bev_f32: channels-last map in one pass) == dense().view(N, C*D, H, W),
    forward and backward, bit for bit (pure data movement)."""
    from unidistill_amd.ops import spconv as sp
    rng = np.random.default_rng(9)
    shape = (3, 2, 21, 19)
    coords = _sites(rng, *shape, 0.3)
    feat = rng.standard_normal((len(coords), 20)).astype(np.float32)
    x = _tensor(coords, feat, shape)
    f1 = x.features.detach().clone().requires_grad_(True)
    bev = x.replace_feature(f1).bev()
    assert bev.dtype == torch.float32 and bev.shape == (3, 40, 21, 19) and bev.is_contiguous(memory_format=torch.channels_last)
    f2 = x.features.detach().clone().requires_grad_(True)
    sp.FUSED_BEV = False
    try:
        ref = x.replace_feature(f2).bev()
    finally:
        sp.FUSED_BEV = True
    assert torch.equal(bev, ref)
    g = torch.randn(3, 40, 21, 19, device="cuda")
    bev.backward(g)
    ref.backward(g)
    assert torch.equal(f1.grad, f2.grad)


@pytest.mark.parametrize("cin,cout", [(64, 64), (128, 64), (32, 32), (32, 64)])
def test_bf16_weight_gradient_many_tiles_per_chunk(cin, cout):
    """The LDS-DMA weight-gradient kernel on a rulebook long enough that every workgroup walks many 64-row
    tiles (index prefetch two tiles ahead, double-buffered DMA, skipped inactive tiles), with the rulebook
    pre-sorted and the gout rows located through row_order (io bit 1), and with nothing permuted: both equal
    the CPU oracle on bf16-rounded operands and each other bit for bit."""
    from unidistill_amd import _lib
    from unidistill_amd.ops import spconv as sp
    rng = np.random.default_rng(cin + cout)
    shape = (1, 12, 70, 80)
    coords = _sites(rng, *shape, 0.35)
    M, K = len(coords), 27
    assert M > 20000
    nbr_np = oracle.spconv_subm_rulebook(coords, shape, (3, 3, 3))
    feat = rng.standard_normal((M, cin)).astype(np.float32)
    gout = rng.standard_normal((M, cout)).astype(np.float32)
    ref = oracle.spconv_wgrad(_bf16(feat), nbr_np, _bf16(gout), cout)
    nbr = torch.from_numpy(nbr_np).cuda()
    f16, g16 = torch.from_numpy(feat).cuda().bfloat16(), torch.from_numpy(gout).cuda().bfloat16()
    lib = _lib.load()
    ws = _lib.workspace(nbr.device, lib.ud_spconv_wgrad_bf16_workspace_bytes(M, K, cin, cout), "spconv_wgrad")
    order = sp.mask_order(nbr, False)
    assert order is not None

    def run(nbr_t, g_t, io, row_order, masks):
        gw = torch.empty((cout, K, cin), dtype=torch.float32, device="cuda")
        _lib.check(lib.ud_spconv_wgrad_bf16(_lib.ptr(f16), _lib.ptr(nbr_t), _lib.ptr(g_t), _lib.ptr(gw), M, K, cin, cout,
                                            io, _lib.ptr(row_order), _lib.ptr(masks), _lib.ptr(ws), ws.numel(),
                                            _lib.stream_of(gw)), "ud_spconv_wgrad_bf16")
        return gw
    a = run(sp.sorted_rulebook(nbr), g16, 3, order, sp.tile_masks(nbr))                       # gout through row_order
    b = run(sp.sorted_rulebook(nbr), g16.index_select(0, order.long()), 1, None, sp.tile_masks(nbr))   # all pre-sorted
    assert torch.equal(a, b)
    np.testing.assert_allclose(a.cpu().numpy(), ref, **_tol(ref))
    c = run(nbr, g16, 1, None, None)                                                          # nothing permuted
    np.testing.assert_allclose(c.cpu().numpy(), ref, **_tol(ref))


@pytest.mark.parametrize("M,K", [(1251, 27), (70000, 27), (5, 3), (4096 * 3 + 17, 8), (400003, 27), (2049, 31), (3000, 1), (513, 10)])
def test_mask_order_vs_numpy(M, K):
    """ud_spconv_mask_order (offset ranks from sampled rows, rarity-weighted masks, stable radix sort) bit for bit against
    its numpy restatement -- index work: exact."""
    from unidistill_amd.ops import spconv as sp
    rng = np.random.default_rng(M + K)
    p = rng.uniform(0.15, 0.95, K)
    nbr_h = np.where(rng.uniform(size=(M, K)) < p[None, :], rng.integers(0, M, (M, K)), -1).astype(np.int32)
    nbr = torch.from_numpy(nbr_h).cuda()
    order = sp.mask_order(nbr, False).cpu().numpy()
    act = nbr_h >= 0
    step = max(1, M // 4096)
    cnt = act[::step].sum(0)
    rank = np.array([sum((cnt[j] > cnt[k]) or (cnt[j] == cnt[k] and j < k) for j in range(K)) for k in range(K)])
    mask = (act.astype(np.int64) << rank[None, :]).sum(1)
    ref = np.argsort(mask, kind="stable")
    assert order.dtype == np.int32 and np.array_equal(order, ref)
    assert sp.mask_order(nbr, True) is sp.mask_order(nbr, False)          # cached on the rulebook, shared with the mirrored pass
