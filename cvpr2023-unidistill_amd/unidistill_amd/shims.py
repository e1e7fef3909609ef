"""Install the MI355X ops under the module names the UNMODIFIED reference imports.

    import unidistill_amd.shims as shims; shims.install()      # before `import unidistill...`

After this, the reference tree's own modules (lss_fpn.py, voxelization.py, spconv_backbone.py,
height_compression.py, the BEVFusion_*_exp files) import and run on this library:

  unidistill.layers.blocks_3d.mmdet3d.voxel_pooling_ext   -> ops.bev_pool (voxel_pooling_forward_wrapper)
  spconv.pytorch / spconv.pytorch.utils / spconv.core     -> ops.spconv, ops.voxelize.PointToVoxel, ConvAlgo
  mmdet.models.build_backbone / mmdet3d.models.build_neck -> layers.image (ResNet-50, SECONDFPN)
  mmcv.Config                                             -> attribute dict
  iou3d_nms_cuda.nms_gpu                                  -> ops.nms.nms_gpu (rotated-BEV NMS, ud_nms_rotated_bev)
  roiaware_pool3d_cuda                                    -> import-time stub (unused path; calling raises)
Registry-style mmdet.core.* names the reference merely imports are provided as inert placeholders.
"""
import sys
import types


class Config(dict):
    """mmcv.Config stand-in: recursive attribute access + dict API (.get/.pop)."""

    def __init__(self, d=None, **kw):
        super().__init__()
        for k, v in dict(d or {}, **kw).items():
            self[k] = v

    def __setitem__(self, k, v):
        if isinstance(v, dict) and not isinstance(v, Config):
            v = Config(v)
        super().__setitem__(k, v)

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    __setattr__ = __setitem__


def _module(name, package=True, **attrs):
    m = sys.modules.get(name)
    if m is None:
        m = types.ModuleType(name)
        if package:
            m.__path__ = []
        sys.modules[name] = m
        if "." in name and package:
            parent, child = name.rsplit(".", 1)
            setattr(_module(parent), child, m)
    for k, v in attrs.items():
        setattr(m, k, v)
    return m


def _leaf(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def _unavailable(what):
    def fn(*a, **k):
        raise NotImplementedError(f"{what} is outside the MI355X hot path (SURVEY.md 8f)")
    return fn


class _Registry:
    def register_module(self, *a, **k):
        return lambda cls: cls


def install():
    from .layers import image
    from .ops import bev_pool, spconv as sp, voxelize
    # --- the reference's three missing pybind extensions (relative imports fall back to sys.modules)
    _leaf("unidistill.layers.blocks_3d.mmdet3d.voxel_pooling_ext",
          voxel_pooling_forward_wrapper=bev_pool.voxel_pooling_forward_wrapper)
    from .ops import nms
    _leaf("unidistill.layers.head.det3d.generate_proposals.iou3d_nms_cuda",
          nms_gpu=nms.nms_gpu, boxes_iou_bev_gpu=nms.boxes_iou_bev_gpu,
          nms_normal_gpu=_unavailable("nms_normal_gpu"))
    _leaf("unidistill.utils.det3d_utils.roiaware_pool3d_cuda",
          **{n: _unavailable("roiaware_pool3d_cuda." + n) for n in (
              "points_in_boxes_cpu", "points_in_boxes_gpu", "bev_in_boxes_cpu", "bev_in_boxes_gpu",
              "points_in_boxes_bev_gpu", "forward", "backward")})
    # --- spconv
    _module("spconv")
    _module("spconv.core", ConvAlgo=sp.ConvAlgo)
    _module("spconv.pytorch", SparseConvTensor=sp.SparseConvTensor, SubMConv3d=sp.SubMConv3d,
            SparseConv3d=sp.SparseConv3d, SparseInverseConv3d=sp.SparseInverseConv3d,
            SparseSequential=sp.SparseSequential, SparseModule=sp.SparseModule)
    _module("spconv.pytorch.utils", PointToVoxel=voxelize.PointToVoxel)
    _module("spconv.pytorch.functional")
    _module("spconv.pytorch.ops")
    # --- mmcv / mmdet / mmdet3d
    _module("mmcv", Config=Config, imnormalize=_unavailable("mmcv.imnormalize (dataset side)"))
    _module("mmdet")
    _module("mmdet.models", build_backbone=image.build_backbone)
    _module("mmdet3d")
    _module("mmdet3d.models", build_neck=image.build_neck)
    _module("mmdet.core")
    _module("mmdet.core.bbox", BaseBBoxCoder=object, AssignResult=object)
    _module("mmdet.core.bbox.assigners", AssignResult=object, BaseAssigner=object)
    _module("mmdet.core.bbox.builder", BBOX_ASSIGNERS=_Registry(), BBOX_CODERS=_Registry())
    _module("mmdet.core.bbox.iou_calculators", build_iou_calculator=_unavailable("iou calculator"))
    _module("mmdet.core.bbox.match_costs", build_match_cost=_unavailable("match cost"))
    _module("mmdet.core.bbox.match_costs.builder", MATCH_COST=_Registry())
    _module("mmdet3d.core")
    _module("mmdet3d.core.bbox")
    _module("mmdet3d.core.bbox.structures")
    _module("mmdet3d.core.bbox.structures.lidar_box3d", LiDARInstance3DBoxes=object)
    return True
