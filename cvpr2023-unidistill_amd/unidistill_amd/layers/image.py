"""Image branch: ResNet-50 backbone + SECONDFPN neck on the HIP convolution / BatchNorm kernels (layers/dense.py); the frozen
7x7 stem + max-pool run on csrc/stem.hip.

The reference builds these through mmdet / mmdet3d (lss_fpn.py:143-149, configs
BEVFusion_nuscenes_centerhead_fusion_exp.py:24-39); neither package nor its source is part of the
reference tree, so these are the standard published architectures (parity unpinned, random
weights in the benchmark).  Parameter names follow mmdet's ResNet (conv1/bn1/layerN.M.convK/bnK/
downsample.0/1) and mmdet3d's SECONDFPN (deblocks.i.0 / deblocks.i.1) so their checkpoints map 1:1.
SURVEY 8f.3 ranks a hand-written NHWC bf16 image branch as "next"; north_star keeps hand-written
MFMA to the BEV trunk + head.
"""
import numpy as np
import torch
from torch import nn

from .dense import Conv2d, ConvTranspose2d, FusedSequential, batchnorm_act
from ..ops import stem as hipstem


def _resolve_checkpoint(spec):
    """mmcv's Pretrained initialiser resolves ``torchvision://resnet50`` through the network; here: a local file,
    the UD_RESNET50_CKPT environment variable, or torch hub's checkpoint cache.  -> path or None."""
    import glob
    import os
    if os.path.isfile(spec):
        return spec
    if spec.startswith("torchvision://"):
        env = os.environ.get("UD_RESNET50_CKPT", "")
        if os.path.isfile(env):
            return env
        name = spec[len("torchvision://"):]
        hits = sorted(glob.glob(os.path.join(torch.hub.get_dir(), "checkpoints", name + "-*.pth")))
        if hits:
            return hits[0]
    return None


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = Conv2d(planes, planes, 3, stride=stride, padding=1, bias=False)   # "pytorch" style
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample

    def forward(self, x):
        # each convolution hands the statistics pass of the training-mode BatchNorm behind it over from its epilogue
        # the identity branch -- x itself or its downsample convolution -- is fed from the alias conv1 hands back, so its gradient
        # joins inside conv1's data-gradient kernel (the same fp32 sum as autograd's separate 3-pass add: 830 MB at layer2.0)
        y, x2 = self.conv1.forward_with_skip(x, self.bn1.training)
        idt = x2 if self.downsample is None else self.downsample(x2)
        y = batchnorm_act(self.bn1, y)
        y = batchnorm_act(self.bn2, self.conv2(y, self.bn2.training))
        return batchnorm_act(self.bn3, self.conv3(y, self.bn3.training), residual=idt)


class ResNet(nn.Module):
    """ResNet-50/101 trunk returning the feature maps listed in ``out_indices``."""
    arch = {50: (3, 4, 6, 3), 101: (3, 4, 23, 3)}
    hip_stem = True            # class-wide switch (tests / A-B timing)

    def __init__(self, depth=50, out_indices=(0, 1, 2, 3), frozen_stages=-1, norm_eval=False,
                 init_cfg=None, **_):
        super().__init__()
        self.out_indices = tuple(out_indices)
        self.frozen_stages, self.norm_eval = frozen_stages, norm_eval
        self.init_cfg = init_cfg
        self.conv1 = nn.Conv2d(3, 64, 7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, stride=2, padding=1)
        inplanes = 64
        for i, n in enumerate(self.arch[depth]):
            planes, stride = 64 * 2 ** i, (1 if i == 0 else 2)
            down = FusedSequential(Conv2d(inplanes, planes * 4, 1, stride=stride, bias=False),
                                   nn.BatchNorm2d(planes * 4))
            blocks = [Bottleneck(inplanes, planes, stride, down)]
            inplanes = planes * 4
            blocks += [Bottleneck(inplanes, planes) for _ in range(1, n)]
            setattr(self, f"layer{i + 1}", nn.Sequential(*blocks))
        self._freeze_stages()

    def _freeze_stages(self):
        """mmdet ResNet._freeze_stages: frozen_stages >= 0 freezes the stem (conv1 / bn1: no gradient,
        bn1 on its running statistics), frozen_stages = n >= 1 additionally layer1..layer_n.  The
        reference configuration has frozen_stages=0 (centerhead_fusion_exp.py:24-31)."""
        if self.frozen_stages >= 0:
            self.bn1.eval()
            for m in (self.conv1, self.bn1):
                for p in m.parameters():
                    p.requires_grad = False
        for i in range(1, self.frozen_stages + 1):
            m = getattr(self, f"layer{i}")
            m.eval()
            for p in m.parameters():
                p.requires_grad = False

    def init_weights(self):
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)
        cfg = self.init_cfg or {}
        self.pretrained_loaded = False
        if cfg.get("type") == "Pretrained":
            import os
            import warnings
            spec = str(cfg.get("checkpoint", ""))
            path = _resolve_checkpoint(spec)
            if path is not None:
                state = torch.load(path, map_location="cpu", weights_only=True)
                state = state.get("state_dict", state)
                missing = self.load_state_dict(state, strict=False)
                if missing.missing_keys:
                    warnings.warn(f"ResNet init_cfg: {len(missing.missing_keys)} keys missing in {path}")
                self.pretrained_loaded = True
            elif os.environ.get("UD_RANDOM_INIT", "0") == "1":
                # benchmark / test mode (synthetic data, random weights of the reference architecture): keep the
                # reference's freeze pattern -- its compute graph -- on the Kaiming initialisation
                warnings.warn(f"ResNet init_cfg checkpoint {spec!r} not found: UD_RANDOM_INIT=1 keeps the Kaiming "
                              "initialisation (benchmark mode; the reference starts from ImageNet weights)")
            elif self.frozen_stages >= 0:
                # a frozen, randomly initialised stem can never be trained: refuse to TRAIN on it -- at the first training
                # forward, not here, so that building the model in order to load a full trained checkpoint (evaluation,
                # resume) works without any environment variable
                self._missing_pretrained = (
                    f"ResNet init_cfg checkpoint {spec!r} not found and frozen_stages={self.frozen_stages} freezes "
                    "the stem: load a checkpoint (load_state_dict), point UD_RESNET50_CKPT (or init_cfg['checkpoint']) at a "
                    "local torchvision ResNet-50 state_dict, or set UD_RANDOM_INIT=1 for synthetic benchmarks / tests")
                self.register_load_state_dict_post_hook(ResNet._checkpoint_loaded)
            else:
                warnings.warn(f"ResNet init_cfg checkpoint {spec!r} not found: Kaiming initialisation kept")

    @staticmethod
    def _checkpoint_loaded(module, incompatible_keys):
        """load_state_dict post-hook: the refusal to train on a random frozen stem is lifted only by a load that actually
        carried this network's weights.  The hook also fires for a PARENT's load (strict=False, a checkpoint holding only the
        LiDAR teacher, ...): if any of this module's own parameters / buffers is among the missing keys, the stem is still random."""
        own = {k for k in module.state_dict().keys() if not k.endswith("num_batches_tracked")}
        missing = incompatible_keys.missing_keys
        if any(m in own or any(m.endswith("." + k) for k in own) for m in missing):
            return
        module._missing_pretrained = None

    def stem_takes_any_layout(self, x):
        """True when the stem runs on the HIP kernel, which reads x through its strides (no layout copy needed in front)."""
        w, mp = self.conv1.weight, self.maxpool
        if not ResNet.hip_stem:
            return False
        return bool(Conv2d.hip_enabled and x.is_cuda and w.is_contiguous(memory_format=torch.channels_last)
                    and not w.is_contiguous() and hipstem.supported(x, self.conv1, self.bn1)
                    and (mp.kernel_size, mp.stride, mp.padding, mp.dilation, mp.ceil_mode) == (3, 2, 1, 1, False))

    def _hip_stem(self, x):
        """The frozen stem on csrc/stem.hip (conv1 + eval bn1 + ReLU in one fp32-MFMA kernel, then the max-pool): NHWC models
        only (the stem filter in channels-last memory is the model's layout flag, see train.to_channels_last)."""
        if not self.stem_takes_any_layout(x):
            return None
        ac = torch.is_autocast_enabled("cuda") and torch.get_autocast_dtype("cuda") == torch.bfloat16
        return hipstem.stem(x, self.conv1, self.bn1, torch.bfloat16 if ac else torch.float32)

    _missing_pretrained = None

    def forward(self, x):
        if self._missing_pretrained and self.training and torch.is_grad_enabled():
            raise FileNotFoundError(self._missing_pretrained)
        y = self._hip_stem(x)
        x = self.maxpool(batchnorm_act(self.bn1, self.conv1(x))) if y is None else y
        outs = []
        for i in range(4):
            x = getattr(self, f"layer{i + 1}")(x)
            if i in self.out_indices:
                outs.append(x)
        return tuple(outs)

    def train(self, mode=True):
        super().train(mode)
        self._freeze_stages()
        if mode and self.norm_eval:
            for m in self.modules():
                if isinstance(m, nn.BatchNorm2d):
                    m.eval()
        return self


class SECONDFPN(nn.Module):
    """Per-level (de)conv to a common stride, BN(eps 1e-3, momentum 0.01) + ReLU, concat."""

    def __init__(self, in_channels, out_channels, upsample_strides, **_):
        super().__init__()
        assert len(in_channels) == len(out_channels) == len(upsample_strides)
        self.deblocks = nn.ModuleList()
        for cin, cout, s in zip(in_channels, out_channels, upsample_strides):
            if s >= 1:
                k = int(s)
                op = ConvTranspose2d(cin, cout, k, stride=k, bias=False)
            else:
                k = int(np.round(1 / s))
                op = Conv2d(cin, cout, k, stride=k, bias=False)
            self.deblocks.append(FusedSequential(op, nn.BatchNorm2d(cout, eps=1e-3, momentum=0.01),
                                                 nn.ReLU(inplace=True)))

    def init_weights(self):
        for m in self.modules():
            if isinstance(m, (nn.Conv2d, nn.ConvTranspose2d)):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")

    def forward(self, feats):
        ups = [blk(f) for blk, f in zip(self.deblocks, feats)]
        return [torch.cat(ups, 1) if len(ups) > 1 else ups[0]]


# type name -> class, like mmdet's BACKBONES / mmdet3d's NECKS registries (lss_fpn.py:143-149 builds through
# them); tests register small stand-in networks here to pin LSSFPN / the model against the reference
BACKBONES = {"ResNet": ResNet}
NECKS = {"SECONDFPN": SECONDFPN}


def build_backbone(cfg):
    cfg = dict(cfg)
    kind = cfg.pop("type")
    if kind not in BACKBONES:
        raise NotImplementedError(f"image backbone {kind!r}: every experiment overrides the default "
                                  "Swin config with ResNet-50 (centerhead_fusion_exp.py:24-31)")
    return BACKBONES[kind](**cfg)


def build_neck(cfg):
    cfg = dict(cfg)
    kind = cfg.pop("type")
    if kind not in NECKS:
        raise NotImplementedError(kind)
    return NECKS[kind](**cfg)
