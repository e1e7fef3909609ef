"""Shrunk BEVFusion configuration + small stand-in image networks shared by the golden generator
(tests/golden/make_goldens.py, which builds the REFERENCE model from it) and the parity tests (which
build this repository's model from it).  Original test code: nothing here comes from the reference.

Geometry: 24 m x 24 m scene, 0.075 m voxels, out_size_factor 8 -> 40 x 40 BEV cells of 0.6 m; two
cameras of 64 x 176 px, stride 16 -> 4 x 11 feature pixels, 24 depth bins of 0.5 m, 8 context channels.
"""
import copy

import torch
from torch import nn

PCR = [-12.0, -12.0, -5.0, 12.0, 12.0, 3.0]
VOXEL = [0.075, 0.075, 0.2]
OSF = 8
GRID = [320, 320, 40]
IMG_DIM = (64, 176)
CLASS_NAMES = ["car", "truck", "bus", "barrier"]
TASKS = [dict(num_class=1, class_names=["car"]), dict(num_class=2, class_names=["truck", "bus"]),
         dict(num_class=1, class_names=["barrier"])]
COMMON_HEADS = {"iou": [1, 2], "reg": [2, 2], "height": [1, 2], "dim": [3, 2], "rot": [2, 2], "vel": [2, 2]}
GEOMETRY = dict(point_cloud_range=PCR, voxel_size=VOXEL, out_size_factor=OSF, grid_size=GRID)


class TinyBackbone(nn.Module):
    """3 -> 6 channels at stride 8, -> 10 channels at stride 16 (conv + BN + ReLU each)."""

    def __init__(self, **_):
        super().__init__()
        self.c1 = nn.Conv2d(3, 6, 8, stride=8, bias=False)
        self.b1 = nn.BatchNorm2d(6)
        self.c2 = nn.Conv2d(6, 10, 3, stride=2, padding=1, bias=False)
        self.b2 = nn.BatchNorm2d(10)

    def init_weights(self):
        pass

    def forward(self, x):
        f8 = torch.relu(self.b1(self.c1(x)))
        f16 = torch.relu(self.b2(self.c2(f8)))
        return f8, f16


class TinyNeck(nn.Module):
    """stride-8 map down to stride 16, concatenated with the stride-16 map: 6 + 10 = 16 channels."""

    def __init__(self, **_):
        super().__init__()
        self.down = nn.Conv2d(6, 6, 2, stride=2)

    def init_weights(self):
        pass

    def forward(self, feats):
        return [torch.cat([self.down(feats[0]), feats[1]], 1)]


CAMERA_ENCODER = dict(
    x_bound=[PCR[0], PCR[3], VOXEL[0] * OSF], y_bound=[PCR[1], PCR[4], VOXEL[1] * OSF],
    z_bound=[PCR[2], PCR[5], PCR[5] - PCR[2]], d_bound=[2.0, 14.0, 0.5], final_dim=IMG_DIM,
    output_channels=8, downsample_factor=16, img_backbone_conf=dict(type="TinyBackbone"),
    img_neck_conf=dict(type="TinyNeck"), depth_net_conf=dict(in_channels=16, mid_channels=16))
BEV_ENCODER = dict(backbone2d_layer_nums=[1, 1], backbone2d_layer_strides=[1, 2], backbone2d_num_filters=[8, 16],
                   backbone2d_upsample_strides=[1, 2], backbone2d_num_upsample_filters=[12, 12],
                   num_bev_features=8, backbone2d_use_scconv=False)
MAX_OBJS = 200


def ours_model_cfg():
    """model_cfg for unidistill_amd.models.BEVFusionCenterHead (camera only)."""
    det = dict(tasks=copy.deepcopy(TASKS), out_size_factor=OSF, max_objs=MAX_OBJS, dense_reg=1, assign_topk=9,
               gaussian_overlap=0.1, min_radius=2, with_velocity=True, input_channels=24, grid_size=GRID,
               point_cloud_range=PCR, voxel_size=VOXEL, code_weights=[1.0] * 8 + [0.2, 0.2], loc_weight=0.25,
               iou_weight=5.0, share_conv_channel=16, common_heads=copy.deepcopy(COMMON_HEADS),
               init_bias=-2.19, focal_alpha=0.25, focal_gamma=2, proposal=None)
    return dict(class_names=CLASS_NAMES, lidar_encoder=None, camera_encoder=copy.deepcopy(CAMERA_ENCODER),
                bev_encoder=copy.deepcopy(BEV_ENCODER), det_head=det)


def reference_model_cfg():
    """The same model in the reference's MODEL_CFG / CENTERPOINT_DET_HEAD_CFG layout
    (base_nuscenes_cfg.py:105-283)."""
    det = dict(
        class_name=CLASS_NAMES,
        target_assigner=dict(
            densehead_out_size_factor=OSF, densehead_tasks=copy.deepcopy(TASKS), target_assigner_dense_reg=1,
            target_assigner_gaussian_overlap=0.1, target_assigner_max_objs=MAX_OBJS, target_assigner_min_radius=2,
            target_assigner_mapping={n: i + 1 for i, n in enumerate(CLASS_NAMES)}, grid_size=GRID,
            pc_range=PCR[0:2], voxel_size=VOXEL[0:2], target_assigner_topk=9, target_assigner_no_log=False,
            with_velocity=True),
        proposal_layer=dict(
            densehead_dataset_name="nuscenes", densehead_tasks=copy.deepcopy(TASKS),
            proposal_post_center_limit_range=[-14.0, -14.0, -10.0, 14.0, 14.0, 10.0], proposal_score_threshold=0.1,
            proposal_pc_range=PCR[0:2], densehead_out_size_factor=OSF, proposal_voxel_size=VOXEL[0:2],
            no_log=False, proposal_iou_aware_list=[0.65] * 4, nms_iou_threshold_train=0.8,
            nms_pre_max_size_train=100, nms_post_max_size_train=20, nms_iou_threshold_test=0.1,
            nms_pre_max_size_test=100, nms_post_max_size_test=20),
        dense_head=dict(
            densehead_dataset_name="nuscenes", densehead_tasks=copy.deepcopy(TASKS), densehead_out_size_factor=OSF,
            input_channels=24, grid_size=GRID, point_cloud_range=PCR,
            densehead_loss_code_weights=[1.0] * 8 + [0.2, 0.2], densehead_loss_loc_weight=0.25,
            densehead_loss_iou_weight=5.0, densehead_share_conv_channel=16,
            densehead_common_heads=copy.deepcopy(COMMON_HEADS), densehead_upsample_for_pedestrian=False,
            densehead_mode="3d", densehead_init_bias=-2.19),
        target_assigner_alpha=0.25, target_assigner_gamma=2)
    return dict(class_names=CLASS_NAMES, lidar_encoder=None, camera_encoder=copy.deepcopy(CAMERA_ENCODER),
                bev_encoder=copy.deepcopy(BEV_ENCODER), det_head=det)
