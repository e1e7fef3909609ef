"""Import harness for the UniDistill reference -- used ONLY by tests/golden/make_goldens.py.

Runs in the build container where /root/reference exists; never on the GPU box.  The reference
needs third-party packages that are not installed (mmcv, mmdet, mmdet3d, spconv, numba,
pytorch_lightning, nuscenes, torchvision, ...) and three prebuilt CUDA extensions that are missing
from its tree.  Minimal ``sys.modules`` stand-ins let its pure-torch modules import and run on CPU
so their outputs can be captured as golden vectors.  Nothing here ships with the product.
"""
import os
import sys
import types

import torch
import torch.nn as nn

REF_ROOT = os.environ.get("UNIDISTILL_REF", "/root/reference")


class _AttrDict(dict):
    """Recursive attribute dict with .get/.pop, standing in for mmcv.Config."""

    def __init__(self, d=None, **kw):
        super().__init__()
        for k, v in dict(d or {}, **kw).items():
            self[k] = self._wrap(v)

    @classmethod
    def _wrap(cls, v):
        if isinstance(v, dict) and not isinstance(v, _AttrDict):
            return cls(v)
        return v

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = self._wrap(v)


def _mod(name, **attrs):
    m = sys.modules.get(name)
    if m is None:
        m = types.ModuleType(name)
        m.__path__ = []  # behave like a package so submodules resolve
        sys.modules[name] = m
        if "." in name:
            parent, child = name.rsplit(".", 1)
            setattr(_mod(parent), child, m)
    for k, v in attrs.items():
        setattr(m, k, v)
    return m


def _leaf(name, **attrs):
    """Register only the leaf module (real parent packages are imported normally later;
    ``from . import leaf`` falls back to sys.modules)."""
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


class _Registry:
    def register_module(self, *a, **k):
        def deco(cls):
            return cls
        return deco


def _identity_decorator(*a, **k):
    if len(a) == 1 and callable(a[0]) and not k:
        return a[0]
    return lambda f: f


class _NoInitModule(nn.Module):
    def init_weights(self):
        pass


def _cpu_voxel_pooling_forward_wrapper(B, N, C, nx, ny, nz, geom, feat, out, pos):
    """CPU stand-in with the call-site contract of lss_fpn.py:48-59 (sequential index_add)."""
    nx, ny, nz = int(nx), int(ny), int(nz)
    g = geom.reshape(B, N, 3).long()
    kept = (g[..., 0] >= 0) & (g[..., 0] < nx) & (g[..., 1] >= 0) & (g[..., 1] < ny) \
        & (g[..., 2] >= 0) & (g[..., 2] < nz)
    o = out.view(B * ny * nx, C)
    for b in range(B):
        k = kept[b]
        idx = (b * ny + g[b, k, 1]) * nx + g[b, k, 0]
        o.index_add_(0, idx, feat[b][k])
        pm = pos[b]
        pm[k, 0] = b
        pm[k, 1] = g[b, k, 1].to(pos.dtype)
        pm[k, 2] = g[b, k, 0].to(pos.dtype)
    return 0


def install(backbone_factory=None, neck_factory=None):
    """Install the stand-ins and put the reference on sys.path.  Idempotent."""
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)
    sys.dont_write_bytecode = True  # the reference tree is read-only
    if not hasattr(torch.Tensor, "_ud_orig_cuda"):
        torch.Tensor._ud_orig_cuda = torch.Tensor.cuda
        torch.Tensor.cuda = lambda self, *a, **k: self  # hard .cuda() calls in the reference
        nn.Module.cuda = lambda self, *a, **k: self

    # missing prebuilt extensions (.MISSING_LARGE_BLOBS)
    _leaf("unidistill.utils.det3d_utils.roiaware_pool3d_cuda")
    _leaf("unidistill.layers.head.det3d.generate_proposals.iou3d_nms_cuda")
    _leaf("unidistill.layers.blocks_3d.mmdet3d.voxel_pooling_ext",
         voxel_pooling_forward_wrapper=_cpu_voxel_pooling_forward_wrapper)

    bf = backbone_factory or (lambda cfg: _NoInitModule())
    nf = neck_factory or (lambda cfg: _NoInitModule())
    _mod("mmdet")
    _mod("mmdet.models", build_backbone=bf)
    _mod("mmdet3d")
    _mod("mmdet3d.models", build_neck=nf)
    _mod("mmdet.core")
    _mod("mmdet.core.bbox", BaseBBoxCoder=object, AssignResult=object)
    _mod("mmdet.core.bbox.assigners", AssignResult=object, BaseAssigner=object)
    _mod("mmdet.core.bbox.builder", BBOX_ASSIGNERS=_Registry(), BBOX_CODERS=_Registry())
    _mod("mmdet.core.bbox.iou_calculators", build_iou_calculator=lambda *a, **k: None)
    _mod("mmdet.core.bbox.match_costs", build_match_cost=lambda *a, **k: None)
    _mod("mmdet.core.bbox.match_costs.builder", MATCH_COST=_Registry())
    _mod("mmdet3d.core")
    _mod("mmdet3d.core.bbox")
    _mod("mmdet3d.core.bbox.structures")
    _mod("mmdet3d.core.bbox.structures.lidar_box3d", LiDARInstance3DBoxes=object)
    _mod("mmcv", Config=_AttrDict, imnormalize=lambda img, *a, **k: img)
    _mod("torchvision")
    _mod("torchvision.ops", roi_align=None)
    _mod("sklearn")
    _mod("sklearn.datasets", load_sample_images=None)

    class LightningModule(nn.Module):
        def save_hyperparameters(self, *a, **k):
            pass
    _mod("pytorch_lightning", LightningModule=LightningModule, Trainer=object,
         seed_everything=lambda *a, **k: None)
    _mod("pytorch_lightning.core", LightningModule=LightningModule)
    _mod("pytorch_lightning.callbacks")
    _mod("numba", jit=_identity_decorator, njit=_identity_decorator)
    _mod("nuscenes")
    for sub in ("nuscenes", "utils", "utils.data_classes", "utils.geometry_utils", "eval",
                "eval.detection", "eval.detection.config", "eval.detection.evaluate",
                "eval.detection.data_classes", "eval.common", "eval.common.loaders",
                "eval.detection.constants", "eval.common.data_classes", "eval.common.utils",
                "eval.detection.algo", "eval.detection.utils", "utils.splits"):
        m = _mod("nuscenes." + sub)
        for n in ("NuScenes", "Box", "LidarPointCloud", "RadarPointCloud", "DetectionEval",
                  "config_factory", "DetectionConfig", "view_points", "transform_matrix",
                  "DetectionBox", "DetectionMetrics", "DetectionMetricDataList", "EvalBoxes",
                  "load_prediction", "load_gt", "add_center_dist", "filter_eval_boxes",
                  "accumulate", "calc_ap", "calc_tp", "TP_METRICS", "create_splits_scenes",
                  "DETECTION_NAMES", "center_distance", "scale_iou", "yaw_diff", "velocity_l2",
                  "attr_acc", "cummean", "DetectionMetricData"):
            setattr(m, n, object)
    _mod("pyquaternion", Quaternion=object)
    _mod("skimage")
    _mod("skimage.io")
    _mod("skimage.transform")
    _mod("cv2")

    class _ConvAlgo:
        Native = 0
        MaskImplicitGemm = 1
    _mod("spconv")
    _mod("spconv.core", ConvAlgo=_ConvAlgo)

    def _no_spconv(*a, **k):
        raise NotImplementedError("spconv is a third-party binary; not available for goldens")
    _mod("spconv.pytorch", SparseModule=nn.Module, SparseSequential=nn.Sequential,
         SubMConv3d=_no_spconv, SparseConv3d=_no_spconv, SparseInverseConv3d=_no_spconv,
         SparseConvTensor=_no_spconv)
    _mod("spconv.pytorch.utils", PointToVoxel=_no_spconv)
    _mod("spconv.pytorch.functional")
    _mod("spconv.pytorch.ops")
    return REF_ROOT
