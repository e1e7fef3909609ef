"""Frozen ResNet stem on the HIP kernels of csrc/stem.hip: conv1 (7x7 / stride 2) + eval-mode bn1 + ReLU in one fp32-MFMA
kernel, max-pool 3x3 / stride 2 in a second one (mmdet ResNet as the reference configures it: frozen_stages = 0,
unidistill/layers/blocks_3d/mmdet3d/lss_fpn.py:143-149).  Forward only: nothing in front of the stem takes a gradient."""
import ctypes

import torch

from .. import _lib
from . import bn_act as hipbn


def supported(x, conv, bn):
    w = conv.weight
    return (x.is_cuda and x.dim() == 4 and x.dtype == torch.float32 and x.shape[1] == 3 and not x.requires_grad
            and tuple(w.shape) == (64, 3, 7, 7) and conv.stride == (2, 2) and conv.padding == (3, 3)
            and conv.dilation == (1, 1) and conv.groups == 1 and conv.bias is None and not w.requires_grad
            and w.dtype == torch.float32 and not bn.training and bn.track_running_stats
            and not (bn.weight is not None and bn.weight.requires_grad))


def _packed(conv):
    """The frozen filters in the kernel's operand order, once per version of the weight tensor."""
    w = conv.weight
    key = (w.data_ptr(), w._version, w.device)
    hit = getattr(conv, "_ud_stem_pack", None)
    if hit is None or hit[0] != key:
        host = w.detach().to("cpu", torch.float32)
        out = torch.empty(7 * 6 * 4 * 16 * 4, dtype=torch.float32)
        sn, sc, sky, skx = host.stride()
        _lib.check(_lib.load().ud_stem_pack_weights(host.data_ptr(), sn, sc, sky, skx, out.data_ptr()),
                   "ud_stem_pack_weights")
        hit = (key, out.to(w.device))
        conv._ud_stem_pack = hit
    return hit[1]


def stem(x, conv, bn, out_dtype=torch.float32, pool=True):
    """maxpool(relu(bn(conv(x)))) -> [B, 64, H', W'] in channels-last memory (``out_dtype`` float32 or bfloat16)."""
    _lib.require_gpu(x)
    if not supported(x, conv, bn):
        raise ValueError("stem(): frozen 7x7 / stride-2 stem with eval-mode BatchNorm on a float32 GPU image batch expected")
    lib = _lib.load()
    B, _, H, W = x.shape
    OH, OW = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    dev = x.device
    vec = hipbn.batch_stats(torch.empty((1, 64, 1, 1), device=dev), bn.weight, bn.bias, bn.running_mean, bn.running_var,
                            False, bn.momentum, bn.eps)
    wpk = _packed(conv)
    bf16 = out_dtype == torch.bfloat16
    y = torch.empty((B, OH, OW, 64), dtype=out_dtype, device=dev)
    sb, sc, sy, sx = x.stride()
    st = _lib.stream_of(x)
    _lib.check(lib.ud_stem_conv7x7_bn_relu(x.data_ptr(), sb, sc, sy, sx, B, H, W, wpk.data_ptr(), vec[3].data_ptr(),
                                           vec[4].data_ptr(), y.data_ptr(), 1 if bf16 else 0, st),
               "ud_stem_conv7x7_bn_relu")
    if not pool:
        return y.permute(0, 3, 1, 2)
    PH, PW = (OH - 1) // 2 + 1, (OW - 1) // 2 + 1
    z = torch.empty((B, PH, PW, 64), dtype=out_dtype, device=dev)
    _lib.check(lib.ud_maxpool3x3s2_nhwc(y.data_ptr(), z.data_ptr(), B, OH, OW, 64, 1 if bf16 else 0, st),
               "ud_maxpool3x3s2_nhwc")
    return z.permute(0, 3, 1, 2)                  # logical NCHW, channels-last memory
