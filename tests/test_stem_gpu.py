"""Frozen ResNet stem (csrc/stem.hip: conv 7x7 / stride 2 + eval BatchNorm + ReLU on the fp32 MFMA pipe, max-pool 3x3 / stride 2)
against plain PyTorch fp32 ops -- the oracle of a floating-point kernel.  Reference call site: the mmdet ResNet built in
unidistill/layers/blocks_3d/mmdet3d/lss_fpn.py:143-149 (frozen_stages = 0)."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _modules(seed):
    torch.manual_seed(seed)
    conv = torch.nn.Conv2d(3, 64, 7, stride=2, padding=3, bias=False).cuda()
    bn = torch.nn.BatchNorm2d(64).cuda().eval()
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5)
        bn.bias.normal_()
        bn.running_mean.normal_()
        bn.running_var.uniform_(0.5, 2.0)
    for p in list(conv.parameters()) + list(bn.parameters()):
        p.requires_grad = False
    return conv, bn


def _ref(x, conv, bn, pool=True):
    with torch.no_grad():
        y = F.relu(bn(F.conv2d(x.double(), conv.weight.double(), None, 2, 3).float()))
        return F.max_pool2d(y, 3, 2, 1) if pool else y


@pytest.mark.parametrize("shape", [(2, 64, 96), (1, 37, 53), (3, 256, 704), (1, 16, 70), (2, 15, 17)])
@pytest.mark.parametrize("layout", ["nchw", "nhwc", "view"])
def test_stem_vs_pytorch_fp32(hip_lib, shape, layout):
    from unidistill_amd.ops import stem
    B, H, W = shape
    conv, bn = _modules(B * 1000 + H)
    x = torch.randn(B, 3, H, W, device="cuda") * 2 + 0.3
    if layout == "nhwc":
        x = x.contiguous(memory_format=torch.channels_last)
    elif layout == "view":                                # a strided view (crop of a larger batch)
        big = torch.randn(B, 3, H + 5, W + 3, device="cuda")
        big[:, :, 2:2 + H, 1:1 + W] = x
        x = big[:, :, 2:2 + H, 1:1 + W]
    for pool in (False, True):
        got = stem.stem(x, conv, bn, torch.float32, pool=pool)
        ref = _ref(x, conv, bn, pool)
        assert got.shape == ref.shape
        tol = 2e-5 * float(ref.abs().max())               # fp32 kernels: 2e-5 of the output's range
        assert float((got - ref).abs().max()) <= tol, (float((got - ref).abs().max()), tol)
        assert got.permute(0, 2, 3, 1).is_contiguous()    # channels-last memory
    # bf16 output = the fp32 result rounded once
    got16 = stem.stem(x, conv, bn, torch.bfloat16, pool=True)
    ref16 = F.max_pool2d(_ref(x, conv, bn, False).bfloat16().float(), 3, 2, 1)
    assert float((got16.float() - ref16).abs().max()) <= 2 ** -7 * float(ref16.abs().max())


def test_stem_nan_and_padding(hip_lib):
    """A NaN pixel reaches exactly the outputs whose windows contain it (ReLU and max-pool keep NaNs, as PyTorch's do); image
    borders are zero-padded for the convolution and ignored by the pool."""
    from unidistill_amd.ops import stem
    conv, bn = _modules(7)
    x = torch.randn(1, 3, 40, 72, device="cuda")
    x[0, 1, 17, 33] = float("nan")
    got = stem.stem(x, conv, bn)
    ref = _ref(x, conv, bn)
    assert torch.equal(torch.isnan(got), torch.isnan(ref))
    ok = ~torch.isnan(ref)
    assert float((got[ok] - ref[ok]).abs().max()) <= 2e-5 * float(ref[ok].abs().max())


def test_resnet_uses_the_hip_stem(hip_lib, monkeypatch):
    """The channels-last ResNet routes its frozen stem through the HIP kernels (no library convolution left in the image branch)
    and gives the library path's features."""
    monkeypatch.setenv("UD_RANDOM_INIT", "1")
    from unidistill_amd.layers import image
    from unidistill_amd.ops import stem
    from unidistill_amd import train
    torch.manual_seed(3)
    net = image.ResNet(depth=50, out_indices=(2, 3), frozen_stages=0, norm_eval=False,
                       init_cfg=dict(type="Pretrained", checkpoint="torchvision://resnet50")).cuda()
    net.init_weights()
    train.to_channels_last(net)
    net.train()
    x = torch.randn(2, 3, 64, 96, device="cuda")
    calls = []
    orig = stem.stem
    monkeypatch.setattr(stem, "stem", lambda *a, **k: (calls.append(1), orig(*a, **k))[1])
    assert net.stem_takes_any_layout(x)
    with torch.no_grad():
        y = net._hip_stem(x)
        ref = net.maxpool(F.relu(net.bn1(F.conv2d(x, net.conv1.weight, None, 2, 3))))
    assert calls and float((y - ref).abs().max()) <= 1e-4 * float(ref.abs().max())
    net.conv1.weight.requires_grad_(True)                 # a trainable stem stays on the autograd path
    assert not net.stem_takes_any_layout(x)
