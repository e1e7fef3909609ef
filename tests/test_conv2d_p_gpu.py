"""The persistent 32x32x16-MFMA 3x3 kernel (csrc/conv2d_p.hip, reached through ud_conv3x3_nhwc_bf16 on maps with at least 384
work units) against torch.nn.functional.conv2d in fp32: both channel widths, full and half tiles, ragged maps, every
epilogue option, the data-gradient mode (reversed taps) and the BatchNorm partial sums."""
import ctypes

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _persistent_kernel_on(hip_lib):
    old = hip_lib.ud_conv3x3_persistent(1)
    yield
    hip_lib.ud_conv3x3_persistent(old)


def _mk(B, Cin, H, W, Cout, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, Cin, H, W, generator=g).bfloat16()
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) / (3 * Cin ** 0.5)).bfloat16().float()
    return x, w, torch.randn(Cout, generator=g)


SHAPES = [(4, 128, 180, 180, 128),     # trunk: 23 half bands (odd), ragged last column tile, 2 slices of Cin
          (2, 64, 90, 90, 256),        # two n tiles, rows 80..89 in a full tile of which 10 rows are valid
          (8, 64, 64, 48, 64),         # narrow (64-channel) workgroups, exact tiles
          (20, 192, 37, 53, 72),       # Cout not a multiple of 64 (clamped weight rows), 3 slices
          (1, 64, 180, 180, 2688),     # the head's first layer: 21 n tiles
          (12, 256, 16, 44, 256)]      # ResNet layer 3 map


@pytest.mark.parametrize("B,Cin,H,W,Cout", SHAPES)
def test_persistent_kernel_forward_and_dgrad(hip_lib, B, Cin, H, W, Cout):
    from unidistill_amd.ops import conv2d as c2
    x, w, b = _mk(B, Cin, H, W, Cout, Cin + Cout + H)
    dev = torch.device("cuda:0")
    xd = x.to(dev).contiguous(memory_format=torch.channels_last)
    wd = w.to(dev)
    ref = F.conv2d(x.float().to(dev), wd, b.to(dev), 1, 1)
    y = c2._launch(xd, c2.tap_major(wd), Cout, bias=b.to(dev))
    tol = 6e-3 * float(ref.abs().max())
    assert float((y.float() - ref).abs().max()) <= tol
    if Cout % 64:
        return
    # data gradient: transposed weights, taps walked in reverse
    gy = torch.randn(B, Cout, H, W, generator=torch.Generator().manual_seed(3)).bfloat16().to(dev)
    gref = F.conv_transpose2d(gy.float(), wd, None, 1, 1)
    gx = c2._launch(gy.contiguous(memory_format=torch.channels_last), c2.tap_major_transposed(wd), Cin, reverse_taps=True)
    assert float((gx.float() - gref).abs().max()) <= 6e-3 * float(gref.abs().max())


def test_persistent_kernel_fused_epilogue_and_stats(hip_lib):
    from unidistill_amd import _lib
    from unidistill_amd.ops import conv2d as c2
    B, Cin, H, W, Cout = 6, 128, 100, 70, 128
    x, w, b = _mk(B, Cin, H, W, Cout, 11)
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(4)
    scale, shift = torch.rand(Cout, generator=g) + 0.5, torch.randn(Cout, generator=g)
    res = torch.randn(B, Cout, H, W, generator=g).bfloat16()
    conv = F.conv2d(x.float(), w, b, 1, 1)
    ref = F.relu(conv * scale[None, :, None, None] + shift[None, :, None, None] + res.float())
    xd = x.to(dev).contiguous(memory_format=torch.channels_last)
    y = c2.conv3x3_inference(xd, w.to(dev), b.to(dev), scale.to(dev), shift.to(dev),
                             res.to(dev).contiguous(memory_format=torch.channels_last), relu=True)
    np.testing.assert_allclose(y.float().cpu().numpy(), ref.numpy(), rtol=0, atol=6e-3 * float(ref.abs().max()))
    # BatchNorm partials: per-unit (sum, sum of squares) of the STORED bf16 values, reduced in slice order
    yb, (part, ns, rows) = c2._launch(xd, c2.tap_major(w.to(dev)), Cout, bias=b.to(dev), bn_stats=True)
    assert rows == B * H * W and ns == B * ((H + 7) // 8) * ((W + 15) // 16)
    tot = part[:ns * Cout * 2].view(ns, Cout, 2).double().sum(0).cpu()
    stored = yb.float().double().cpu()
    np.testing.assert_allclose(tot[:, 0].numpy(), stored.sum((0, 2, 3)).numpy(), rtol=1e-5, atol=1e-2)
    np.testing.assert_allclose(tot[:, 1].numpy(), stored.square().sum((0, 2, 3)).numpy(), rtol=1e-5, atol=1e-2)
    y2, _ = c2._launch(xd, c2.tap_major(w.to(dev)), Cout, bias=b.to(dev), bn_stats=True)
    assert torch.equal(yb, y2)                                        # deterministic


def test_dispatch_switch(hip_lib):
    assert hip_lib.ud_conv3x3_persistent(-1) == 1          # the fixture turned it on
    assert hip_lib.ud_conv3x3_persistent(0) == 1 and hip_lib.ud_conv3x3_persistent(1) == 0
