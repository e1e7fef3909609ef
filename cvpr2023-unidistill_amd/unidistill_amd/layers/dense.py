"""Dense 2-D building blocks that route 3x3 convolutions to the hand-written MFMA kernel.

``Conv2d`` is nn.Conv2d (same parameters, same state_dict keys); in the bf16 mixed-precision mode on
the GPU its 3x3 / stride-1 / pad-1 case runs ops/conv2d.py (ud_conv3x3_nhwc_bf16) for the forward
and the data gradient.  ``FusedSequential`` is nn.Sequential that additionally
  * folds ``ZeroPad2d(1)`` + unpadded 3x3 conv (the first conv of every BaseBEVBackbone level,
    reference base_bev_backbone.py:48-58) into the kernel's implicit padding, and
  * in inference (no autograd, BatchNorm in eval) runs conv + BatchNorm2d + ReLU as ONE kernel
    through the fused epilogue -- the frozen distillation teacher's trunk.
Anything else (fp32 mode, CPU, other kernel sizes / strides) is the plain PyTorch module.
"""
import os

import torch
from torch import nn

from .. import _lib
from ..ops import bn_act as hipbn, conv2d as hipconv, conv2d_f32 as hipconv32
from ..ops.spconv import folded_batchnorm, wants_grad


def _mixed_precision(x):
    return x.is_cuda and (x.dtype == torch.bfloat16 or (
        torch.is_autocast_enabled("cuda") and torch.get_autocast_dtype("cuda") == torch.bfloat16))


def _is3x3(conv, padding):
    return (isinstance(conv, nn.Conv2d) and conv.kernel_size == (3, 3) and conv.stride == (1, 1)
            and conv.padding == (padding, padding) and conv.dilation == (1, 1) and conv.groups == 1
            and conv.padding_mode == "zeros" and conv.in_channels % 64 == 0
            and conv.out_channels % 64 == 0)


def _is3x3_any(conv, padding):
    return (isinstance(conv, nn.Conv2d) and conv.kernel_size == (3, 3) and conv.stride == (1, 1)
            and conv.padding == (padding, padding) and conv.dilation == (1, 1) and conv.groups == 1
            and conv.padding_mode == "zeros")


def _is3x3_s2(conv, padding):
    return (isinstance(conv, nn.Conv2d) and conv.kernel_size == (3, 3) and conv.stride == (2, 2)
            and conv.padding == (padding, padding) and conv.dilation == (1, 1) and conv.groups == 1
            and conv.padding_mode == "zeros" and conv.bias is None)


def _is1x1(conv):
    return (isinstance(conv, nn.Conv2d) and conv.kernel_size == (1, 1) and conv.stride == (1, 1)
            and conv.padding == (0, 0) and conv.dilation == (1, 1) and conv.groups == 1)


def _fp32_mode(x):
    return x.is_cuda and x.dtype == torch.float32 and not torch.is_autocast_enabled("cuda")


def _fp32_kernel_pays(conv, x):
    """fp32 mode.  3x3: the straight-line tap kernel with per-launch tile heights measures 103-124 TFLOP/s against
    MIOpen's 48-108 on every shape of the step but two small ResNet maps where the two are within 3 % (tools/
    time_conv2d_f32.py), so all stride-1 3x3 convolutions run on it.  1x1: channel counts the library pads (head / depth
    net: 2.1x) and every map of at least 1 k pixels (forward + data gradient of a layer together: 0.93-1.07x of the library,
    94.6 vs 94.9 ms per step); on the 16 x 44 / 8 x 22 ResNet maps the library is ~1.1x faster per layer, which does not show
    in the step (89.5 vs 90.1 ms), so the default is "all": every convolution of the fp32 step but the frozen 7 x 7 stem is
    hand-written and deterministic; UD_HIP_FP32_CONV=pays restores the per-shape routing."""
    if Conv2d.hip_fp32 == "all":
        return True
    px = x.shape[2] * x.shape[3]
    if conv.kernel_size == (3, 3):
        return True
    return conv.out_channels % 64 != 0 or px >= 1024


# fp32 mode: strided / transposed convolutions on the fp32 mapped kernels (forward, data and weight gradient)
_FP32_MAPPED = os.environ.get("UD_FP32_MAPPED", "1") != "0"


def _fp32_mapped(x):
    return _FP32_MAPPED and Conv2d.hip_fp32 and _fp32_mode(x)


class Conv2d(nn.Conv2d):
    hip_enabled = True          # class-wide switch (tests / A-B timing)
    # fp32 MFMA kernels for the fp32 (reference) mode: False / True (where they pay) / "all"
    hip_fp32 = {"0": False, "pays": True}.get(os.environ.get("UD_HIP_FP32_CONV", "all"), "all")

    def forward(self, x, bn_stats=False):
        """bn_stats: the caller feeds the result straight into a training-mode BatchNorm (batchnorm_act): the hand-written
        kernels then emit the BatchNorm's per-tile sums from their epilogue (one pass over the output saved)."""
        if Conv2d.hip_enabled and x.dim() == 4 and _mixed_precision(x):
            if _is3x3(self, 1):
                return hipconv.conv3x3(x.to(torch.bfloat16), self.weight, self.bias, bn_stats)
            if self._hip_1x1(x):
                return hipconv.conv1x1(x.to(torch.bfloat16), self.weight, self.bias, bn_stats)
            y = self._hip_permutation_conv(x)
            if y is not None:
                return y
        elif Conv2d.hip_enabled and Conv2d.hip_fp32 and x.dim() == 4 and _fp32_mode(x):
            if _is3x3_any(self, 1) and hipconv32.supported(x, self.weight, 3) and _fp32_kernel_pays(self, x):
                return hipconv32.conv3x3(x, self.weight, self.bias, bn_stats)
            if _is1x1(self) and hipconv32.supported(x, self.weight, 1) and _fp32_kernel_pays(self, x):
                return hipconv32.conv1x1(x, self.weight, self.bias, bn_stats)
            if _FP32_MAPPED:
                y = self._hip_permutation_conv(x)       # strided convolutions: fp32 twin of the mapped 1x1 kernel
                if y is not None:
                    return y
        if Conv2d.hip_enabled:
            _lib.library_fallthrough("layers.dense.Conv2d", x, self.weight, kernel=self.kernel_size, stride=self.stride,
                                     padding=self.padding)
        return super().forward(x)

    def _hip_permutation_conv(self, x):
        """k = s / stride s (image-neck levels) and 1x1 / stride s (ResNet stage shortcuts): im2col is a permutation,
        so all three passes run on the 1x1 MFMA kernels through a pixel map (ops/conv2d.py)."""
        k, st = self.kernel_size, self.stride
        if _is3x3_s2(self, 1) and hipconv.supported_3x3_s2(x, self.weight):
            y = hipconv.conv3x3_stride2(x, self.weight)         # ResNet stages 2-4, first block
            return y if self.bias is None else y + self.bias.to(y.dtype).view(1, -1, 1, 1)
        if not (k[0] == k[1] and st[0] == st[1] and st[0] >= 2 and self.padding == (0, 0) and self.dilation == (1, 1)
                and self.groups == 1 and self.padding_mode == "zeros"):
            return None
        s_ = st[0]
        if k[0] == s_ and hipconv.supported_patch(x, self.weight, s_):
            y = hipconv.conv_patch(x, self.weight, s_)
        elif k[0] == 1 and hipconv.supported_1x1(x, self.weight, strided=True):
            y = hipconv.conv1x1_strided(x, self.weight, s_)
        else:
            return None
        return y if self.bias is None else y + self.bias.to(y.dtype).view(1, -1, 1, 1)

    def _hip_1x1(self, x):
        return (_is1x1(self) and self.in_channels % 64 == 0 and self.out_channels % 8 == 0
                and x.is_contiguous(memory_format=torch.channels_last))

    def forward_with_skip(self, x, bn_stats=False):
        """(self(x), x'): x' is x, handed back through the convolution's autograd node so that the gradient of
        an identity branch fed from it is added inside the data-gradient kernel (residual blocks)."""
        if (Conv2d.hip_enabled and x.dim() == 4 and _mixed_precision(x) and x.dtype == torch.bfloat16
                and self._hip_1x1(x) and torch.is_grad_enabled() and x.requires_grad):
            return hipconv.conv1x1_skip(x, self.weight, self.bias, bn_stats)
        if (Conv2d.hip_enabled and Conv2d.hip_fp32 and x.dim() == 4 and _fp32_mode(x) and _is1x1(self)
                and hipconv32.supported(x, self.weight, 1) and self.weight.shape[0] % 32 == 0
                and _fp32_kernel_pays(self, x) and torch.is_grad_enabled() and x.requires_grad):
            return hipconv32.conv1x1_skip(x, self.weight, self.bias, bn_stats)
        return self(x, bn_stats), x


class ConvTranspose2d(nn.ConvTranspose2d):
    """nn.ConvTranspose2d (same parameters / state_dict keys).  Its 1x1 / stride-1 case -- the first deblock
    of BaseBEVBackbone and of the image neck (reference base_bev_backbone.py:67-92, upsample stride 1) -- is
    a plain 1x1 convolution with the weight's two channel axes swapped and runs on hipconv.conv1x1."""

    def forward(self, x, output_size=None):
        if (Conv2d.hip_enabled and output_size is None and x.dim() == 4 and _mixed_precision(x)
                and self.kernel_size == (1, 1) and self.stride == (1, 1) and self.padding == (0, 0)
                and self.output_padding == (0, 0) and self.dilation == (1, 1) and self.groups == 1
                and self.in_channels % 64 == 0 and self.out_channels % 8 == 0
                and x.is_contiguous(memory_format=torch.channels_last)):
            return hipconv.conv1x1(x.to(torch.bfloat16), self.weight.permute(1, 0, 2, 3), self.bias)
        if (Conv2d.hip_enabled and Conv2d.hip_fp32 and output_size is None and x.dim() == 4 and _fp32_mode(x)
                and self.kernel_size == (1, 1) and self.stride == (1, 1) and self.padding == (0, 0)
                and self.output_padding == (0, 0) and self.dilation == (1, 1) and self.groups == 1
                and hipconv32.supported(x, self.weight.permute(1, 0, 2, 3), 1)):
            # fp32 mode: the same 1x1 convolution on the fp32 MFMA kernels (all three passes)
            return hipconv32.conv1x1(x, self.weight.permute(1, 0, 2, 3), self.bias)
        if (Conv2d.hip_enabled and output_size is None and x.dim() == 4 and (_mixed_precision(x) or _fp32_mapped(x))
                and self.kernel_size[0] == self.kernel_size[1] == self.stride[0] == self.stride[1] and self.stride[0] >= 2
                and self.padding == (0, 0) and self.output_padding == (0, 0) and self.dilation == (1, 1)
                and self.groups == 1 and hipconv.supported_patch(x, self.weight, self.stride[0], transposed=True)):
            # k = s / stride s: every input pixel writes its own s x s output block (BaseBEVBackbone's up-sampling
            # deblock, base_bev_backbone.py:67-92; the image neck's last level) -- the 1x1 kernels with an output map
            y = hipconv.conv_transpose_patch(x, self.weight, self.stride[0])
            return y if self.bias is None else y + self.bias.to(y.dtype).view(1, -1, 1, 1)
        if Conv2d.hip_enabled:
            _lib.library_fallthrough("layers.dense.ConvTranspose2d", x, self.weight, kernel=self.kernel_size, stride=self.stride)
        return super().forward(x, output_size)


def _can_fuse_inference(x, bn, conv):
    """conv + BN + ReLU as one inference kernel: BN in eval and no gradient wanted by the input, the
    conv or the BN (a frozen-BN fine-tune keeps trainable conv weights: that must take the autograd path)."""
    return (isinstance(bn, nn.BatchNorm2d) and not bn.training and bn.affine and bn.track_running_stats
            and not wants_grad(x, bn, conv))


_HIP_BN = os.environ.get("UD_HIP_BN", "1") != "0"      # A/B switch for the streaming BatchNorm kernels


def batchnorm_act(bn, x, residual=None, relu=True, out=None):
    """relu(bn(x) (+ residual)): the streaming HIP kernels in bf16 channels-last mode (training-mode
    statistics with autograd, or eval-mode without), the PyTorch ops otherwise.  out: a channel slice of a concatenation
    buffer (ops/bn_act.cat_buffer) the HIP path writes in place; the other paths ignore it (the caller checks the result)."""
    frozen_grad = (not bn.training) and wants_grad(x, residual, bn)
    if Conv2d.hip_enabled and _HIP_BN and isinstance(bn, (nn.BatchNorm2d, nn.BatchNorm1d)) \
            and not frozen_grad and hipbn.supported(x, bn):
        if out is not None and not (x.dim() == 4 and out.shape == x.shape and out.dtype == x.dtype):
            out = None
        return hipbn.bn_act(bn, x, residual, relu, out)
    if Conv2d.hip_enabled and _HIP_BN and not frozen_grad:
        _lib.library_fallthrough("layers.dense.batchnorm_act", x, training=bn.training)
    y = bn(x)
    if residual is not None:
        y = y + residual
    return torch.relu(y) if relu else y


def _feeds_training_bn(mods, j):
    """mods[j] is a BatchNorm2d in training mode: the convolution in front of it can hand over its statistics."""
    return j < len(mods) and isinstance(mods[j], nn.BatchNorm2d) and mods[j].training and _HIP_BN


class FusedSequential(nn.Sequential):
    def forward(self, x, out=None):
        """out: where a trailing BatchNorm (+ ReLU) may write its result (see batchnorm_act)."""
        mods = list(self)
        i, n = 0, len(mods)
        while i < n:
            m = mods[i]
            conv, skip = None, 0
            if Conv2d.hip_enabled and x.dim() == 4 and _mixed_precision(x):
                if isinstance(m, nn.ZeroPad2d) and tuple(m.padding) == (1, 1, 1, 1) and i + 1 < n \
                        and _is3x3(mods[i + 1], 0):
                    conv, skip = mods[i + 1], 2
                elif _is3x3(m, 1):
                    conv, skip = m, 1
            if conv is None and Conv2d.hip_enabled and x.dim() == 4 and (_mixed_precision(x) or _fp32_mapped(x)) \
                    and isinstance(m, nn.ZeroPad2d) and tuple(m.padding) == (1, 1, 1, 1) and i + 1 < n \
                    and _is3x3_s2(mods[i + 1], 0) and hipconv.supported_3x3_s2(x, mods[i + 1].weight):
                # ZeroPad2d(1) + unpadded 3x3 / stride 2 (first conv of BaseBEVBackbone's second level) == pad-1 conv
                x = hipconv.conv3x3_stride2(x, mods[i + 1].weight)
                i += 2
                continue
            if conv is None and Conv2d.hip_enabled and Conv2d.hip_fp32 and x.dim() == 4 and _fp32_mode(x) \
                    and isinstance(m, nn.ZeroPad2d) and tuple(m.padding) == (1, 1, 1, 1) and i + 1 < n \
                    and _is3x3_any(mods[i + 1], 0) and hipconv32.supported(x, mods[i + 1].weight, 3) \
                    and _fp32_kernel_pays(mods[i + 1], x):
                # fp32 mode: ZeroPad2d(1) + unpadded 3x3 conv == the kernel's implicit padding
                j = i + 2
                if j < n and _can_fuse_inference(x, mods[j], mods[i + 1]):      # frozen network: conv + BN (+ ReLU) in one kernel
                    relu = j + 1 < n and isinstance(mods[j + 1], nn.ReLU)
                    scale, shift = folded_batchnorm(mods[j])
                    x = hipconv32.conv3x3_inference(x, mods[i + 1].weight, mods[i + 1].bias, scale, shift, relu)
                    i = j + (2 if relu else 1)
                    continue
                x = hipconv32.conv3x3(x, mods[i + 1].weight, mods[i + 1].bias, _feeds_training_bn(mods, i + 2))
                i += 2
                continue
            if conv is None:
                if isinstance(m, nn.BatchNorm2d) and x.dim() == 4:
                    relu = i + 1 < n and isinstance(mods[i + 1], nn.ReLU)
                    last = i + (2 if relu else 1) >= n
                    x = batchnorm_act(m, x, None, relu, out if last else None)
                    i += 2 if relu else 1
                    continue
                if (isinstance(m, Conv2d) and Conv2d.hip_enabled and Conv2d.hip_fp32 and x.dim() == 4 and _fp32_mode(x)
                        and _is3x3_any(m, 1) and hipconv32.supported(x, m.weight, 3) and _fp32_kernel_pays(m, x)
                        and i + 1 < n and _can_fuse_inference(x, mods[i + 1], m)):
                    relu = i + 2 < n and isinstance(mods[i + 2], nn.ReLU)       # frozen network, fp32: conv + BN (+ ReLU) fused
                    scale, shift = folded_batchnorm(mods[i + 1])
                    x = hipconv32.conv3x3_inference(x, m.weight, m.bias, scale, shift, relu)
                    i += 3 if relu else 2
                    continue
                x = m(x, _feeds_training_bn(mods, i + 1)) if isinstance(m, Conv2d) else m(x)
                i += 1
                continue
            j = i + skip
            bn = mods[j] if j < n else None
            if bn is not None and _can_fuse_inference(x, bn, conv):
                relu = j + 1 < n and isinstance(mods[j + 1], nn.ReLU)
                scale, shift = folded_batchnorm(bn)
                x = hipconv.conv3x3_inference(x.to(torch.bfloat16), conv.weight, conv.bias, scale, shift,
                                              None, relu)
                i = j + (2 if relu else 1)
            else:
                x = hipconv.conv3x3(x.to(torch.bfloat16), conv.weight, conv.bias, _feeds_training_bn(mods, j))
                i = j
        return x
