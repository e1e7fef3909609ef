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


# ---- camera side + collate ---------------------------------------------------------------------
IMG_MEAN, IMG_STD, TO_RGB = (123.675, 116.28, 103.53), (58.395, 57.12, 57.375), True    # base_nuscenes_cfg.py:31


def image_normalize(imgs_u8, mean=IMG_MEAN, std=IMG_STD, to_rgb=TO_RGB, channels_last=False):
    """ImageNormalize.forward (transforms3d.py:350-368 -> mmcv.imnormalize) + the dataset's HWC -> CHW permute /
    stack (nuscenes_multimodal.py:262-293) for uint8 images on the device: imgs_u8 [..., H, W, 3] -> float32
    [..., 3, H, W] (``channels_last``: same shape, NHWC memory)."""
    _lib.require_gpu(imgs_u8)
    if imgs_u8.dtype != torch.uint8 or imgs_u8.shape[-1] != 3 or imgs_u8.dim() < 3:
        raise ValueError("imgs_u8 must be uint8 [..., H, W, 3]")
    x = imgs_u8.contiguous()
    lead, (H, W) = x.shape[:-3], x.shape[-3:-1]
    NI = int(np.prod(lead)) if lead else 1
    import ctypes
    f3 = lambda v: (ctypes.c_float * 3)(*[float(a) for a in v])
    if channels_last:
        out = torch.empty((NI, H, W, 3), dtype=torch.float32, device=x.device)
    else:
        out = torch.empty((NI, 3, H, W), dtype=torch.float32, device=x.device)
    _lib.check(_lib.load().ud_image_normalize(_lib.ptr(x), _lib.ptr(out), f3(mean), f3(std), 1 if to_rgb else 0,
                                              NI, H, W, 1 if channels_last else 0, _lib.stream_of(x)),
               "ud_image_normalize")
    if channels_last:
        out = out.permute(0, 3, 1, 2)              # [NI, 3, H, W] view of the NHWC buffer
    return out.reshape(*lead, 3, H, W)             # only splits the leading axis: a view in both layouts


def _fill_batch_tensor(batch_data, device):
    """fill_batch_tensor of collate_fn (nuscenes_multimodal.py:441-463): stack equal-length samples, zero-pad
    ragged ones to the longest (one ud_collate_pad launch); float32 on ``device``."""
    ts = [d if torch.is_tensor(d) else torch.as_tensor(np.asarray(d)) for d in batch_data]
    lens = [len(t) for t in ts]
    if max(lens) == min(lens):
        return torch.stack([t.to(device=device, dtype=torch.float32, non_blocking=True) for t in ts])
    tail = next(tuple(t.shape[1:]) for t in ts if t.numel() != 0)
    W = int(np.prod(tail)) if tail else 1
    L, B = max(lens), len(ts)
    dev_ts = [t.to(device=device, dtype=torch.float32, non_blocking=True).contiguous() for t in ts]
    out = torch.empty((B, L) + tail, dtype=torch.float32, device=device)
    import ctypes
    ptrs = (ctypes.c_void_p * B)(*[t.data_ptr() if t.numel() else None for t in dev_ts])
    rows = (ctypes.c_int64 * B)(*[n if t.numel() else 0 for n, t in zip(lens, dev_ts)])
    _lib.check(_lib.load().ud_collate_pad(ptrs, rows, B, L, W, _lib.ptr(out), _lib.stream_of(out)), "ud_collate_pad")
    return out


def collate_fn(data, device="cuda", is_return_depth=False, with_points=True):
    """collate_fn of the reference (data/multisensorfusion/nuscenes_multimodal.py:418-495) with the batch
    assembled ON THE DEVICE: same keys, shapes and dtypes (float32) -- ``imgs`` [B, sweeps, cams, 3, h, w],
    ``points`` [B, Nmax, D] zero padded, ``gt_boxes`` [B, Mmax, S], ``gt_labels`` [B, Mmax], ``mats_dict`` of
    stacked 4x4 matrices, ``img_metas`` passed through.  A sample may carry ``imgs_u8`` ([sweeps, cams, H, W, 3]
    uint8, not yet normalised) instead of ``imgs``: normalisation + permute then run here in one launch."""
    device = torch.device(device)
    batch = {}
    if "imgs_u8" in data[0]:
        u8 = torch.stack([torch.as_tensor(np.asarray(d["imgs_u8"])) for d in data]).to(device, non_blocking=True)
        batch["imgs"] = image_normalize(u8)
    for key in ("imgs", "points", "gt_boxes", "gt_labels"):
        if key in data[0] and key not in batch:
            batch[key] = _fill_batch_tensor([d[key] for d in data], device)
    if "mats_dict" in data[0]:
        batch["mats_dict"] = {}
        for key in ("sensor2ego_mats", "intrin_mats", "ida_mats", "sensor2sensor_mats", "bda_mat"):
            if key in data[0]["mats_dict"]:
                batch["mats_dict"][key] = torch.stack(
                    [torch.as_tensor(np.asarray(d["mats_dict"][key])) if not torch.is_tensor(d["mats_dict"][key])
                     else d["mats_dict"][key] for d in data]).to(device=device, dtype=torch.float32)
    batch["img_metas"] = [d.get("img_metas") for d in data]
    return batch
