"""LiDAR input side on the device: sweep collection and the BEV augmentation of points / boxes.

Mirrors the reference's numpy transforms (unidistill/data/multisensorfusion/transforms3d.py:379-443,
functional.py:595-646): the 4x4 matrices are built on the host in float64 exactly as the reference builds
them (they are a handful of flops), the per-point work runs in ONE ud_points_transform launch per batch.
"""
import numpy as np
import torch

from .. import _lib


def points_transform(points, seg, mats, last=None, out=None):
    """points f32 [rows, D]; seg: row offsets of the S segments (S+1 ints); mats: [S,4,4] float64; last:
    optional per-segment value for the last column (NaN = keep).  Returns the transformed cloud."""
    _lib.require_gpu(points)
    if points.dtype != torch.float32 or points.dim() != 2 or points.shape[1] < 3 or not points.is_contiguous():
        raise ValueError("points must be a contiguous float32 [rows, D >= 3] tensor")
    seg = [int(v) for v in seg]
    S = len(seg) - 1
    if seg[0] != 0 or seg[-1] != points.shape[0] or any(b < a for a, b in zip(seg, seg[1:])):
        raise ValueError("seg must ascend from 0 to the number of rows")
    dev = points.device
    mats_d = torch.as_tensor(np.ascontiguousarray(np.asarray(mats, np.float64).reshape(S, 16)), device=dev)
    seg_d = torch.tensor(seg, dtype=torch.int64, device=dev)
    last_d = None if last is None else torch.as_tensor(np.asarray(last, np.float32).reshape(S), device=dev)
    out = torch.empty_like(points) if out is None else out
    max_rows = max((b - a for a, b in zip(seg, seg[1:])), default=0)
    _lib.check(_lib.load().ud_points_transform(_lib.ptr(points), _lib.ptr(out), _lib.ptr(seg_d), _lib.ptr(mats_d),
                                               _lib.ptr(last_d), S, points.shape[1], max_rows, _lib.stream_of(points)),
               "ud_points_transform")
    return out


def sweep_to_key_matrix(key_lidar_to_ego, key_ego_to_global, sweep_pose):
    """transforms3d.py:394-400 (left-associative product, float64)."""
    L, G, S = (np.asarray(m, np.float64) for m in (key_lidar_to_ego, key_ego_to_global, sweep_pose))
    return np.linalg.inv(L) @ np.linalg.inv(G) @ S @ L


def collect_lidar_sweeps(points, sweep_points, info):
    """CollectLidarSweeps.forward for device clouds: ``points`` [N,D] and the list ``sweep_points``, ``info``
    as in the reference's data_dict["info"] (ego_to_global, lidar_to_ego, timestamp, sweep_lidar_infos).
    Returns the concatenated [N + sum(Ni), D] cloud; for D == 5 the last column is the time lag in seconds."""
    clouds = [points] + list(sweep_points)
    D = points.shape[1]
    mats = [np.eye(4)] + [sweep_to_key_matrix(info["lidar_to_ego"], info["ego_to_global"], s["sweep_lidar_to_ego"])
                          for s in info["sweep_lidar_infos"]]
    nan = float("nan")
    if D == 5:
        last = [0.0] + [(info["timestamp"] - s["sweep_lidar_timestamp"]) / 1e6 for s in info["sweep_lidar_infos"]]
    else:
        last = [nan] * len(clouds)
    seg = np.cumsum([0] + [c.shape[0] for c in clouds])
    return points_transform(torch.cat(clouds).contiguous(), seg, np.stack(mats), last)


def bev_transform_matrix(rotate_deg, scale, trans, flip_dx, flip_dy):
    """functional.bev_transform's matrix (functional.py:595-632), float64."""
    a = rotate_deg / 180 * np.pi
    s, c = np.sin(a), np.cos(a)
    rot = np.array([[c, -s, 0, 0], [s, c, 0, 0], [0, 0, 1, 0], [0, 0, 0, 1]])
    sc = np.diag([scale, scale, scale, 1.0])
    tr = np.eye(4)
    tr[:3, 3] = trans
    flip = np.eye(4)
    if flip_dx:
        flip = flip @ np.diag([-1.0, 1.0, 1.0, 1.0])
    if flip_dy:
        flip = flip @ np.diag([1.0, -1.0, 1.0, 1.0])
    return flip @ tr @ sc @ rot


def bev_affine(points, gt_boxes, rotate_deg, scale, trans, flip_dx, flip_dy):
    """BevAffineTransformation.forward with the drawn augmentation given: points [N,D] and gt_boxes [M,7|9]
    (float32, device) -> (points', gt_boxes', bda_mat float64 4x4).  Box arithmetic follows
    functional.py:633-646 in the reference's precisions (centres in float64, the rest in float32)."""
    rotate_deg, scale = float(rotate_deg), float(scale)
    mat = bev_transform_matrix(rotate_deg, scale, trans, flip_dx, flip_dy)
    out = points_transform(points, [0, points.shape[0]], mat[None])
    boxes = gt_boxes.clone()
    if boxes.shape[0] > 0:
        boxes[:, :7] = points_transform(boxes[:, :7].contiguous(), [0, boxes.shape[0]], mat[None])
        boxes[:, 3:6] = gt_boxes[:, 3:6] * scale
        yaw = gt_boxes[:, 6] + rotate_deg / 180 * np.pi
        if flip_dx:
            yaw = np.pi - yaw
        if flip_dy:
            yaw = -yaw
        boxes[:, 6] = yaw
        if boxes.shape[1] > 7:          # velocities through the 2x2 block: float64 products, one rounding to float32
            v = gt_boxes[:, 7:9].double()
            boxes[:, 7] = (float(mat[0, 0]) * v[:, 0] + float(mat[0, 1]) * v[:, 1]).float()
            boxes[:, 8] = (float(mat[1, 0]) * v[:, 0] + float(mat[1, 1]) * v[:, 1]).float()
    return out, boxes, mat
