"""Synthetic nuScenes-shaped inputs: LiDAR clouds, 6-camera rigs, GT boxes.

All draws come from numpy PCG64 with seed = 1234 + rank so every rank/bench run is reproducible.
Shapes follow the reference's collate_fn contract (data/multisensorfusion/nuscenes_multimodal.py
:418-495): points f32[N,5] (x,y,z,intensity,dt), mats f32[B,1,ncam,4,4], bda f32[B,4,4],
gt_boxes f32[B,M,9], gt_labels f32[B,M].
"""
import numpy as np

POINT_CLOUD_RANGE = (-54.0, -54.0, -5.0, 54.0, 54.0, 3.0)
VOXEL_SIZE = (0.075, 0.075, 0.2)
GRID_SIZE = (1440, 1440, 40)
IMG_DIM = (256, 704)
OUT_SIZE_FACTOR = 8
CAM_YAWS_DEG = (180.0, 250.0, 110.0, 0.0, 55.0, -55.0)


def rng(seed=1234, rank=0):
    return np.random.Generator(np.random.PCG64(seed + rank))


def lidar_cloud(g, n_per_sweep=30000, sweeps=1):
    """f32[N,5]; range-limited ring pattern, roughly what ObjectRangeFilter leaves."""
    out = []
    for s in range(sweeps):
        n = n_per_sweep
        th = g.uniform(0.0, 2 * np.pi, n)
        r = np.minimum(np.abs(g.normal(0.0, 22.0, n)) + 1.0, 75.0)
        z = np.clip(g.normal(-1.2, 0.9, n), -4.9, 2.9)
        x, y = r * np.cos(th), r * np.sin(th)
        keep = (np.abs(x) < 54.0) & (np.abs(y) < 54.0)
        pts = np.stack([x, y, z, g.uniform(0, 255, n), np.full(n, s * 0.05)], 1)[keep]
        out.append(pts)
    return np.concatenate(out, 0).astype(np.float32)


def lidar_cloud_uniform(g, n):
    """Worst-case occupancy: uniform in the detection box."""
    lo, hi = np.array(POINT_CLOUD_RANGE[:3]), np.array(POINT_CLOUD_RANGE[3:])
    xyz = g.uniform(lo + 1e-3, hi - 1e-3, (n, 3))
    return np.concatenate([xyz, g.uniform(0, 255, (n, 1)), np.zeros((n, 1))], 1).astype(np.float32)


def pad_clouds(clouds):
    """collate_fn padding: zero rows up to the longest cloud (nuscenes_multimodal.py:441-463)."""
    n = max(c.shape[0] for c in clouds)
    out = np.zeros((len(clouds), n, clouds[0].shape[1]), np.float32)
    for i, c in enumerate(clouds):
        out[i, :c.shape[0]] = c
    return out


def camera_rig(g, B=1, ncam=6, bda_aug=False, jitter=0.0):
    """sensor2ego, intrin, ida: f32[B,1,ncam,4,4]; bda f32[B,4,4]."""
    s2e = np.zeros((B, 1, ncam, 4, 4), np.float32)
    intr = np.zeros_like(s2e)
    ida = np.zeros_like(s2e)
    base = np.array([[0, 0, 1], [-1, 0, 0], [0, -1, 0]], np.float64)
    for b in range(B):
        for c in range(ncam):
            yaw = np.deg2rad(CAM_YAWS_DEG[c % 6]) + (g.normal(0, jitter) if jitter else 0.0)
            cy, sy = np.cos(yaw), np.sin(yaw)
            rz = np.array([[cy, -sy, 0], [sy, cy, 0], [0, 0, 1]])
            m = np.eye(4)
            m[:3, :3] = rz @ base
            m[:3, 3] = (1.5 * cy, 0.45 * sy, 1.5)
            s2e[b, 0, c] = m
            k = np.eye(4)
            k[0, 0] = k[1, 1] = 1266.4
            k[0, 2], k[1, 2] = 816.3, 491.5
            intr[b, 0, c] = k
            a = np.eye(4)
            a[0, 0] = a[1, 1] = 0.44
            a[1, 3] = -(900 * 0.44 - IMG_DIM[0])
            ida[b, 0, c] = a
    bda = np.tile(np.eye(4, dtype=np.float32), (B, 1, 1))
    if bda_aug:
        th, s = np.deg2rad(17.0), 1.05
        r = np.eye(4)
        r[:2, :2] = s * np.array([[np.cos(th), -np.sin(th)], [np.sin(th), np.cos(th)]])
        r[2, 2] = s
        r[0] *= -1
        bda[:] = r.astype(np.float32)
    return s2e, intr, ida, bda


CLASS_DIMS = np.array([[4.6, 1.95, 1.7], [6.9, 2.5, 2.8], [6.4, 2.8, 3.2], [11.0, 2.9, 3.4],
                       [12.3, 2.9, 3.9], [0.5, 2.5, 1.0], [2.1, 0.8, 1.5], [1.7, 0.6, 1.3],
                       [0.7, 0.7, 1.8], [0.4, 0.4, 1.1]])


def gt_boxes(g, B=1, M=40, Mmax=None):
    """gt_boxes f32[B,Mmax,9] (x,y,z,dx,dy,dz,yaw,vx,vy), gt_labels f32[B,Mmax] (0-based)."""
    Mmax = Mmax or M
    boxes = np.zeros((B, Mmax, 9), np.float32)
    labels = np.zeros((B, Mmax), np.float32)
    for b in range(B):
        lab = g.integers(0, 10, M)
        boxes[b, :M, 0:2] = g.uniform(-50, 50, (M, 2))
        boxes[b, :M, 2] = g.normal(-1.0, 0.5, M)
        boxes[b, :M, 3:6] = CLASS_DIMS[lab] * g.uniform(0.9, 1.1, (M, 3))
        boxes[b, :M, 6] = g.uniform(-np.pi, np.pi, M)
        boxes[b, :M, 7:9] = g.normal(0, 2, (M, 2))
        labels[b, :M] = lab
    return boxes, labels


def frustum_bins_torch(s2e, intr, ida, bda, device, final_dim=IMG_DIM, down=16,
                       d_bound=(2.0, 58.0, 0.5), lo=(-54.0, -54.0, -5.0), size=(0.6, 0.6, 8.0)):
    """Plain-torch frustum -> ego -> BEV bin indices i32[B, ncam*D*fH*fW, 3] (setup/plumbing only;
    the product path is ud_lss_geometry).  Same math as lss_fpn.py:173-240,311-313."""
    import torch
    H, W = final_dim
    fH, fW = H // down, W // down
    d = torch.arange(*d_bound, dtype=torch.float32, device=device)
    u = torch.linspace(0, W - 1, fW, device=device)
    v = torch.linspace(0, H - 1, fH, device=device)
    D = d.numel()
    fr = torch.stack([u.view(1, 1, fW).expand(D, fH, fW), v.view(1, fH, 1).expand(D, fH, fW),
                      d.view(D, 1, 1).expand(D, fH, fW), torch.ones(D, fH, fW, device=device)], -1)
    s2e, intr, ida, bda = (torch.as_tensor(t, device=device) for t in (s2e, intr, ida, bda))
    s2e, intr, ida = s2e[:, 0], intr[:, 0], ida[:, 0]
    B, ncam = s2e.shape[:2]
    p = torch.linalg.inv(ida).view(B, ncam, 1, 1, 1, 4, 4) @ fr.unsqueeze(-1)
    p = torch.cat([p[..., :2, :] * p[..., 2:3, :], p[..., 2:, :]], -2)
    p = (s2e @ torch.linalg.inv(intr)).view(B, ncam, 1, 1, 1, 4, 4) @ p
    p = (bda.view(B, 1, 1, 1, 1, 4, 4) @ p).squeeze(-1)[..., :3]
    size_t = torch.tensor(size, device=device)
    coord = torch.tensor([l + s / 2.0 for l, s in zip(lo, size)], device=device)
    bins = ((p - (coord - size_t / 2.0)) / size_t).int()
    return bins.reshape(B, -1, 3).contiguous(), p
