"""BEVFusion detector (camera / LiDAR / fusion) with the CenterPoint IoU-aware head.

Mirrors, with state_dict-compatible sub-module names,
  BEVFusion / BEVFusionCenterHead / CameraEncoder / DetHead
      unidistill/exps/multisensor_fusion/nuscenes/BEVFusion/BEVFusion_nuscenes_base_exp.py:88-104,
      :164-259;  BEVFusion_nuscenes_centerhead_fusion_exp.py:44-171
  BaseMultiSensorFusion.with_* properties   unidistill/models/multisensor_fusion/base.py:12-40
"""
import torch
from torch import nn

from .layers.bev import BevEncoder, FusionEncoder
from .layers.center_head import CenterHeadIouAware, FCOSAssigner
from .layers.gen_proposals import IouAwareGenProposals
from .layers.lidar import LidarEncoder
from .layers.lss_fpn import LSSFPN
from .ops.distill import feature_tap


class CameraEncoder(nn.Module):
    def __init__(self, camera_encoder_cfg, **kw):
        super().__init__()
        self.backbone = LSSFPN(**camera_encoder_cfg, **kw)

    def forward(self, imgs, mats_dict, is_return_depth=False):
        return self.backbone(imgs, mats_dict, is_return_depth=is_return_depth)


class DetHead(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.det_head_cfg = cfg
        names = [n for t in cfg["tasks"] for n in t["class_names"]]
        assigner = FCOSAssigner(
            out_size_factor=cfg["out_size_factor"], tasks=cfg["tasks"], dense_reg=cfg["dense_reg"],
            gaussian_overlap=cfg["gaussian_overlap"], max_objs=cfg["max_objs"],
            min_radius=cfg["min_radius"], mapping={n: i + 1 for i, n in enumerate(names)},
            grid_size=cfg["grid_size"], pc_range=cfg["point_cloud_range"][0:2],
            voxel_size=cfg["voxel_size"][0:2], assign_topk=cfg["assign_topk"],
            with_velocity=cfg["with_velocity"])
        prop = cfg.get("proposal")
        proposal_layer = None
        if prop is not None:      # BEVFusion_nuscenes_centerhead_fusion_exp.py:67-84
            proposal_layer = IouAwareGenProposals(
                dataset_name="nuscenes", class_names=[t["class_names"] for t in cfg["tasks"]],
                post_center_limit_range=prop["post_center_limit_range"], score_threshold=prop["score_threshold"],
                pc_range=cfg["point_cloud_range"][0:2], out_size_factor=cfg["out_size_factor"],
                voxel_size=cfg["voxel_size"][0:2], no_log=prop["no_log"], iou_aware_list=prop["iou_aware_list"],
                nms_iou_threshold_train=prop["nms_iou_threshold_train"],
                nms_pre_max_size_train=prop["nms_pre_max_size_train"],
                nms_post_max_size_train=prop["nms_post_max_size_train"],
                nms_iou_threshold_test=prop["nms_iou_threshold_test"],
                nms_pre_max_size_test=prop["nms_pre_max_size_test"],
                nms_post_max_size_test=prop["nms_post_max_size_test"])
        self.dense_head = CenterHeadIouAware(
            dataset_name="nuscenes", tasks=cfg["tasks"], target_assigner=assigner, proposal_layer=proposal_layer,
            out_size_factor=cfg["out_size_factor"], input_channels=cfg["input_channels"],
            grid_size=cfg["grid_size"], point_cloud_range=cfg["point_cloud_range"],
            code_weights=cfg["code_weights"], loc_weight=cfg["loc_weight"], iou_weight=cfg["iou_weight"],
            share_conv_channel=cfg["share_conv_channel"], common_heads=cfg["common_heads"],
            init_bias=cfg["init_bias"], focal_alpha=cfg["focal_alpha"], focal_gamma=cfg["focal_gamma"],
            voxel_size_xy=cfg["voxel_size"][0:2])

    def forward(self, x, gt_boxes, targets=None):
        ret = self.dense_head(x, gt_boxes, targets=targets)
        # precomputed targets are cleaned by whoever built them and say so (train.DistillStep.prep); anything else is cleaned here
        if self.training and "box_encoding" in ret and not (targets is not None and targets.get("box_encoding_clean")):
            for enc in ret["box_encoding"].values():
                enc[torch.isinf(enc)] = 0          # log(0) of zero-size boxes (fusion_exp.py:124-126)
        return ret


class BEVFusionCenterHead(nn.Module):
    """forward(lidar_points, cameras_imgs, metas, gt_boxes, return_feature=False)

    training:            (ret_dict{'loss'}, tb_dict, bev_feat, trunk_out, multi_head_features, {})
    return_feature=True: (bev_feat, trunk_out, multi_head_features)
    eval:                the proposal layer's dict: pred_dicts (boxes / scores / labels per sample), rois,
                         roi_scores, roi_labels (layers/gen_proposals.py, rotated NMS on the HIP library)
    """

    def __init__(self, model_cfg, camera_kwargs=None):
        super().__init__()
        self.cfg = model_cfg
        self.class_names = model_cfg["class_names"]
        self.num_class = len(self.class_names)
        self.lidar_encoder = LidarEncoder(model_cfg["lidar_encoder"]) if model_cfg.get("lidar_encoder") else None
        self.camera_encoder = (CameraEncoder(model_cfg["camera_encoder"], **(camera_kwargs or {}))
                               if model_cfg.get("camera_encoder") else None)
        both = self.lidar_encoder is not None and self.camera_encoder is not None
        self.fusion_encoder = FusionEncoder(use_elementwise=False) if both else None
        self.bev_encoder = BevEncoder(model_cfg["bev_encoder"])
        self.det_head = DetHead(model_cfg["det_head"])

    with_lidar_encoder = property(lambda self: self.lidar_encoder is not None)
    with_camera_encoder = property(lambda self: self.camera_encoder is not None)
    with_fusion_encoder = property(lambda self: self.fusion_encoder is not None)

    def extract_bev(self, lidar_points, cameras_imgs, metas, lidar_prepared=None):
        lidar_out = camera_out = None
        if self.with_lidar_encoder:
            lidar_out = self.lidar_encoder(lidar_points, lidar_prepared)
        if self.with_camera_encoder:
            camera_out = self.camera_encoder(cameras_imgs, metas)
        if self.with_fusion_encoder:
            return self.fusion_encoder(lidar_out, camera_out)
        return camera_out if camera_out is not None else lidar_out

    def forward(self, lidar_points=None, cameras_imgs=None, metas=None, gt_boxes=None,
                return_feature=False, targets=None, loss_norm=None, lidar_prepared=None, **_):
        """targets / loss_norm: optional precomputed FCOS targets and globally reduced loss
        normalisers (train.py computes them up front so the network pass holds no collective);
        lidar_prepared: LidarEncoder.prepare(lidar_points), when the caller ran it ahead of time."""
        bev = self.extract_bev(lidar_points, cameras_imgs, metas, lidar_prepared)
        tapped = self.training and not return_feature
        # the two maps the box distillation losses read: their gradients join the network branch's inside one kernel (ops/distill.py)
        bev, bev_out = feature_tap(bev) if tapped else (bev, bev)
        trunk, _ = self.bev_encoder(bev)
        trunk, trunk_out = feature_tap(trunk) if tapped else (trunk, trunk)
        ret = self.det_head(trunk, gt_boxes, targets=targets)
        if return_feature:
            return bev, trunk, ret["multi_head_features"]
        if self.training:
            loss, tb = self.det_head.dense_head.get_loss(ret, norm=loss_norm)
            tb["loss_rpn"] = loss.detach()
            return {"loss": loss}, tb, bev_out, trunk_out, ret["multi_head_features"], {}
        return ret
