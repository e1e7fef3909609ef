"""Shapes and hyper-parameters of the nuScenes BEVFusion / UniDistill experiments.

Values restate unidistill/exps/multisensor_fusion/nuscenes/_base_/base_nuscenes_cfg.py:1-283 and
the overrides of BEVFusion_nuscenes_centerhead_fusion_exp.py:24-41,189-197 (ResNet-50 + SECONDFPN,
depth_net in_channels 512, CenterPoint head).
"""
import copy

POINT_CLOUD_RANGE = [-54.0, -54.0, -5.0, 54.0, 54.0, 3.0]
VOXEL_SIZE = [0.075, 0.075, 0.2]
GRID_SIZE = [1440, 1440, 40]
IMG_DIM = (256, 704)
OUT_SIZE_FACTOR = 8

CLASS_NAMES = ["car", "truck", "construction_vehicle", "bus", "trailer", "barrier", "motorcycle",
               "bicycle", "pedestrian", "traffic_cone"]

DENSE_TASKS = [
    dict(num_class=1, class_names=["car"]),
    dict(num_class=2, class_names=["truck", "construction_vehicle"]),
    dict(num_class=2, class_names=["bus", "trailer"]),
    dict(num_class=1, class_names=["barrier"]),
    dict(num_class=2, class_names=["motorcycle", "bicycle"]),
    dict(num_class=2, class_names=["pedestrian", "traffic_cone"]),
]

LIDAR_ENCODER = dict(point_cloud_range=POINT_CLOUD_RANGE, voxel_size=VOXEL_SIZE, grid_size=GRID_SIZE,
                     max_num_points=10, max_voxels=(120000, 160000), src_num_point_features=5,
                     use_num_point_features=5, map_to_bev_num_features=256)

CAMERA_ENCODER = dict(
    x_bound=[POINT_CLOUD_RANGE[0], POINT_CLOUD_RANGE[3], VOXEL_SIZE[0] * OUT_SIZE_FACTOR],
    y_bound=[POINT_CLOUD_RANGE[1], POINT_CLOUD_RANGE[4], VOXEL_SIZE[1] * OUT_SIZE_FACTOR],
    z_bound=[POINT_CLOUD_RANGE[2], POINT_CLOUD_RANGE[5], POINT_CLOUD_RANGE[5] - POINT_CLOUD_RANGE[2]],
    d_bound=[2.0, 58.0, 0.5], final_dim=IMG_DIM, output_channels=256, downsample_factor=16,
    img_backbone_conf=dict(type="ResNet", depth=50, frozen_stages=0, out_indices=[0, 1, 2, 3],
                           norm_eval=False,
                           init_cfg=dict(type="Pretrained", checkpoint="torchvision://resnet50")),
    img_neck_conf=dict(type="SECONDFPN", in_channels=[256, 512, 1024, 2048],
                       upsample_strides=[0.25, 0.5, 1, 2], out_channels=[128, 128, 128, 128]),
    depth_net_conf=dict(in_channels=512, mid_channels=512))

BEV_ENCODER = dict(backbone2d_layer_nums=[5, 5], backbone2d_layer_strides=[1, 2],
                   backbone2d_num_filters=[128, 256], backbone2d_upsample_strides=[1, 2],
                   backbone2d_num_upsample_filters=[256, 256], num_bev_features=256,
                   backbone2d_use_scconv=False)

DET_HEAD = dict(
    tasks=DENSE_TASKS, out_size_factor=OUT_SIZE_FACTOR, max_objs=2500, dense_reg=1, assign_topk=9,
    gaussian_overlap=0.1, min_radius=2, with_velocity=True, input_channels=512, grid_size=GRID_SIZE,
    point_cloud_range=POINT_CLOUD_RANGE, voxel_size=VOXEL_SIZE,
    code_weights=[1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 0.2, 0.2], loc_weight=0.25, iou_weight=5.0,
    share_conv_channel=64,
    common_heads={"iou": [1, 2], "reg": [2, 2], "height": [1, 2], "dim": [3, 2], "rot": [2, 2],
                  "vel": [2, 2]},
    init_bias=-2.19, focal_alpha=0.25, focal_gamma=2,
    # proposal layer (base_nuscenes_cfg.py:239-255): eval-time decode + rotated NMS
    proposal=dict(post_center_limit_range=[-61.2, -61.2, -10.0, 61.2, 61.2, 10.0], score_threshold=0.1,
                  no_log=False, iou_aware_list=[0.65] * 10,
                  nms_iou_threshold_train=0.8, nms_pre_max_size_train=1500, nms_post_max_size_train=80,
                  nms_iou_threshold_test=0.1, nms_pre_max_size_test=1500, nms_post_max_size_test=100))


def model_cfg(lidar=True, camera=True):
    cfg = dict(class_names=CLASS_NAMES, lidar_encoder=copy.deepcopy(LIDAR_ENCODER) if lidar else None,
               camera_encoder=copy.deepcopy(CAMERA_ENCODER) if camera else None,
               bev_encoder=copy.deepcopy(BEV_ENCODER), det_head=copy.deepcopy(DET_HEAD))
    return cfg


# loss weights / sigmoid clamp / teacher reload of the four distillation experiments
DISTILL_EXPERIMENTS = {
    "camera_exp_distill_lidar": dict(student="camera", teacher="lidar", feat=100.0, rel=40.0, resp=10.0, clamp=1e-4),
    "camera_exp_distill_fusion": dict(student="camera", teacher="fusion", feat=10.0, rel=5.0, resp=10.0, clamp=1e-3),
    "lidar_exp_distill_fusion": dict(student="lidar", teacher="fusion", feat=10.0, rel=1.0, resp=10.0, clamp=1e-4),
    "lidar_exp_distill_camera": dict(student="lidar", teacher="camera", feat=10.0, rel=5.0, resp=1.0, clamp=1e-4),
}
