"""Dense BEV trunk: SECOND-style two-level 2-D backbone with upsampling heads.

State_dict-compatible mirror of BaseBEVBackbone
(unidistill/layers/blocks_2d/det3d/base_bev_backbone.py:10-174) for the configuration the
experiments use (no SC-Conv, base_nuscenes_cfg.py:166-174).  The module layout keeps the reference's
Sequential indices (ZeroPad2d at 0, conv at 1, BN at 2, ...) so checkpoints load unchanged.
In the bf16 mixed-precision mode the 3x3 stride-1 convs run on the hand-written MFMA kernel
(layers/dense.py, ops/conv2d.py); strided / transposed convs and the fp32 mode use PyTorch-ROCm.
"""
import os

import numpy as np
import torch
from torch import nn

from .dense import Conv2d, ConvTranspose2d, FusedSequential
from ..ops import bn_act as hipbn


def _bn(c):
    return nn.BatchNorm2d(c, eps=1e-3, momentum=0.01)


class BaseBEVBackbone(nn.Module):
    fuse_cat = os.environ.get("UD_FUSE_CAT", "1") == "1"      # A/B switch (tests compare the two)

    def __init__(self, layer_nums, layer_strides, num_filters, upsample_strides,
                 num_upsample_filters, input_channels, use_scconv=False, upsample_output=False):
        super().__init__()
        if use_scconv:
            raise NotImplementedError("SC-Conv is config-disabled in every experiment "
                                      "(base_nuscenes_cfg.py:173) and is out of scope")
        layer_nums = list(layer_nums or [])
        layer_strides = list(layer_strides or [])
        num_filters = list(num_filters or [])
        upsample_strides = list(upsample_strides or [])
        num_upsample_filters = list(num_upsample_filters or [])
        assert len(layer_nums) == len(layer_strides) == len(num_filters)
        assert len(upsample_strides) == len(num_upsample_filters)
        c_in = [input_channels] + num_filters[:-1]
        self.blocks = nn.ModuleList()
        self.deblocks = nn.ModuleList()
        for lvl, (n, s, c) in enumerate(zip(layer_nums, layer_strides, num_filters)):
            seq = [nn.ZeroPad2d(1), Conv2d(c_in[lvl], c, 3, stride=s, padding=0, bias=False),
                   _bn(c), nn.ReLU()]
            for _ in range(n):
                seq += [Conv2d(c, c, 3, padding=1, bias=False), _bn(c), nn.ReLU()]
            self.blocks.append(FusedSequential(*seq))
            if upsample_strides:
                us, uc = upsample_strides[lvl], num_upsample_filters[lvl]
                if us >= 1:
                    up = ConvTranspose2d(c, uc, us, stride=us, bias=False)
                else:
                    k = int(np.round(1 / us))
                    up = Conv2d(c, uc, k, stride=k, bias=False)
                self.deblocks.append(FusedSequential(up, _bn(uc), nn.ReLU()))
        c_out = sum(num_upsample_filters)
        if len(upsample_strides) > len(layer_nums):
            us = upsample_strides[-1]
            self.deblocks.append(nn.Sequential(nn.ConvTranspose2d(c_out, c_out, us, stride=us, bias=False),
                                               _bn(c_out), nn.ReLU()))
        self.num_bev_features = c_out
        self.upsample_featuremap = upsample_output
        if upsample_output:
            self.upsample_conv = nn.Sequential(nn.ConvTranspose2d(c_out, c_out, 2, stride=2, bias=False),
                                               _bn(c_out), nn.ReLU())

    def forward(self, spatial_features):
        feats, pyramid = [], {}
        x = spatial_features
        # The upsampling heads write their BatchNorm + ReLU outputs straight into the concatenated map (ud_bn_act_fwd_ld) and
        # read its gradient in place: no cat kernel, no strided-slice copies (530 MB / step at 4 x 512 x 180 x 180 in fp32).
        buf = views = None
        fuse = self.fuse_cat and len(self.deblocks) >= len(self.blocks) > 1 and x.is_cuda and \
            all(isinstance(d, FusedSequential) and len(d) == 3 and isinstance(d[1], nn.BatchNorm2d) for d in self.deblocks[:len(self.blocks)])
        for lvl, block in enumerate(self.blocks):
            x = block(x)
            pyramid["spatial_features_%dx" % int(spatial_features.shape[2] / x.shape[2])] = x
            if not fuse:
                feats.append(self.deblocks[lvl](x) if len(self.deblocks) > 0 else x)
                continue
            if buf is None:
                up = self.deblocks[lvl][0]
                st = up.stride[0] if isinstance(up.stride, tuple) else up.stride
                hw = (x.shape[2] * st, x.shape[3] * st) if isinstance(up, nn.ConvTranspose2d) else (x.shape[2] // st, x.shape[3] // st)
                ref = x.new_empty((x.shape[0], 1, hw[0], hw[1]))
                buf, views = hipbn.cat_buffer(ref, [d[1].num_features for d in self.deblocks[:len(self.blocks)]])
            feats.append(self.deblocks[lvl](x, views[lvl]))          # (a shape / dtype / path mismatch leaves the slice unwritten)
        if fuse and all(f.data_ptr() == v.data_ptr() and f.stride() == v.stride() for f, v in zip(feats, views)):
            x = hipbn.cat_slices(buf, feats)
        else:
            x = torch.cat(feats, dim=1) if len(feats) > 1 else feats[0]
        if len(self.deblocks) > len(self.blocks):
            x = self.deblocks[-1](x)
        if self.upsample_featuremap:
            x = self.upsample_conv(x)
        return x, pyramid


class BevEncoder(nn.Module):
    """BevEncoder of BEVFusion_nuscenes_base_exp.py:138-161 (cfg keys backbone2d_*)."""

    def __init__(self, cfg):
        super().__init__()
        g = cfg.get if hasattr(cfg, "get") else (lambda k, d=None: getattr(cfg, k, d))
        self.bev_encoder_cfg = cfg
        self.backbone_2d = BaseBEVBackbone(
            layer_nums=g("backbone2d_layer_nums"), layer_strides=g("backbone2d_layer_strides"),
            num_filters=g("backbone2d_num_filters"), upsample_strides=g("backbone2d_upsample_strides"),
            num_upsample_filters=g("backbone2d_num_upsample_filters"),
            input_channels=g("num_bev_features"), use_scconv=g("backbone2d_use_scconv", False),
            upsample_output=g("backbone2d_upsample_output", False))

    def forward(self, x):
        return self.backbone_2d(x)


class FusionEncoder(nn.Module):
    """Camera/LiDAR BEV fusion (BEVFusion_nuscenes_base_exp.py:107-135): channel attention over
    the concatenated maps, then 3x3 conv 512 -> 256 + BN + ReLU (or plain sum)."""

    def __init__(self, use_elementwise=True, input_channel=512, output_channel=256, reduction=2):
        super().__init__()
        self.use_elementwise = use_elementwise
        if not use_elementwise:
            self.att = nn.Sequential(nn.AdaptiveAvgPool2d(1),
                                     Conv2d(input_channel, input_channel, 1), nn.Sigmoid())
            self.reduce_conv = FusedSequential(Conv2d(input_channel, output_channel, 3, padding=1, bias=False),
                                             nn.BatchNorm2d(output_channel), nn.ReLU(True))

    def forward(self, x1, x2):
        assert x1.shape == x2.shape, f"shape: {x1.shape} != {x2.shape}"
        if self.use_elementwise:
            return x1 + x2
        x = torch.cat((x1, x2), dim=1)
        return self.reduce_conv(x * self.att(x))
