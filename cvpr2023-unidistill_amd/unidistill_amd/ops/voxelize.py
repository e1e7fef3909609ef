"""LiDAR voxelization + MeanVFE: spconv's ``PointToVoxel`` boundary and the reference modules
built on it.

Reference: unidistill/data/det3d/preprocess/voxelization.py:8-73 (Voxelization),
unidistill/layers/blocks_3d/det3d/vfe/mean_vfe.py:6-34 (MeanVFE).
"""
import ctypes

import numpy as np
import torch
from torch import nn

from .. import _lib


def _f3(v):
    return (ctypes.c_float * len(v))(*[float(x) for x in v])


ALGO = None      # None: by size (below), hash fallback on overflow; 0 / 1 / 2: force one algorithm (tests, timing)
SMALL_CLOUD = 160000   # points: below this the atomic hash (3 launches, ~15 us of atomics) beats the 6-launch partition path
SMALL_TILES = 256      # algo 2 (three launches, self-cleaning workspace): at most this many 1 024-point tiles, P <= 16, B <= 64

# algo 2 leaves its workspace in the clean state its next call starts from.  The note -- the sizes of the last algo-2/3 call -- is
# an ATTRIBUTE OF THE WORKSPACE TENSOR (`_ud_clean`), so a re-allocated (grown) workspace, or a block the caching allocator hands out
# again at the same address to another scope, can never inherit it.  Any other algorithm on the buffer, an exception, or an
# overflow drops it (the next small call then pays the memset); a deferred call's note stays `_ud_clean_pending` until the caller
# has seen m_out[B + 1] == 0 (voxelize_confirm).
def _small_algo(ws, B, N, P, max_voxels):
    """2 (memset first) or 3 (the workspace is known clean) for the three-launch small-cloud path, or None if it does not apply."""
    if B * N >= SMALL_CLOUD or (B * N + 1023) // 1024 > SMALL_TILES or P > 16 or B > 64:
        return None
    return 3 if getattr(ws, "_ud_clean", None) == (B, N, P, max_voxels) else 2


def _forget(ws):
    ws._ud_clean = None
    ws._ud_clean_pending = None


def _note_algo(ws, algo, B, N, P, max_voxels, pending=False):
    _forget(ws)
    if algo in (2, 3):
        setattr(ws, "_ud_clean_pending" if pending else "_ud_clean", (B, N, P, max_voxels))


def _device_workspaces(device):
    idx = device.index if device.index is not None else torch.cuda.current_device()
    return [buf for (dev, _scope, slot), buf in _lib._workspaces.items() if dev == idx and slot == "voxelize"]


def voxelize_batch(points, voxel_size, pc_range, max_points, max_voxels, want_voxels=True,
                   want_mean=True, algo=None):
    """points f32[B,N,F] (cuda) -> (voxels[M,P,F] | None, coords i32[M,4] (b,z,y,x), num i32[M],
    mean f32[M,F] | None, per_sample i32[B]).  One host read of the voxel count sizes the views
    (spconv's PointToVoxel has the same sync); the same read carries the fast path's overflow word, and an
    overflowing input (thousands of points in one voxel) is voxelized again by the hash path."""
    _lib.require_gpu(points)
    if points.dtype != torch.float32:
        raise TypeError("points must be float32")
    if points.dim() == 2:
        points = points.unsqueeze(0)
    points = points.contiguous()
    B, N, F = points.shape
    lib = _lib.load()
    cap = lib.ud_voxelize_capacity(B, N, int(max_voxels))
    need = lib.ud_voxelize_workspace_bytes(B, N, int(max_points), int(max_voxels))
    if cap <= 0 or need == 0:
        raise ValueError("invalid voxelization sizes")
    dev = points.device
    ws = _lib.workspace(dev, need, "voxelize")
    voxels = torch.empty((cap, max_points, F), dtype=torch.float32, device=dev) if want_voxels else None
    coords = torch.empty((cap, 4), dtype=torch.int32, device=dev)
    num = torch.empty((cap,), dtype=torch.int32, device=dev)
    mean = torch.empty((cap, F), dtype=torch.float32, device=dev) if want_mean else None
    m_out = torch.empty((B + 2,), dtype=torch.int32, device=dev)
    algo = ALGO if algo is None else algo
    small = _small_algo(ws, B, N, int(max_points), int(max_voxels))
    if algo == 2 and small is None:
        raise ValueError("ud_voxelize: algo 2 takes at most 256 tiles of 1 024 points, P <= 16, B <= 64")
    order = (algo if algo != 2 else small,) if algo is not None else ((small, 2) if small is not None else
                                                                      ((1,) if B * N < SMALL_CLOUD else (0, 1)))
    for a in order:
        _forget(ws)                                          # unknown until the call has returned
        _lib.check(lib.ud_voxelize(_lib.ptr(points), B, N, F, _f3(voxel_size), _f3(pc_range),
                                   int(max_points), int(max_voxels), _lib.ptr(voxels), _lib.ptr(coords),
                                   _lib.ptr(num), _lib.ptr(mean), _lib.ptr(m_out), _lib.ptr(ws),
                                   ws.numel(), a, _lib.stream_of(points)), "ud_voxelize")
        m_host = m_out.cpu()
        if int(m_host[B + 1]) == 0:
            _note_algo(ws, a, B, N, int(max_points), int(max_voxels))
            break
    else:
        raise RuntimeError("ud_voxelize: a hash partition overflowed (algo 0 forced) or the workspace state was refused")
    M = int(m_host[B])
    return (voxels[:M] if want_voxels else None, coords[:M], num[:M],
            mean[:M] if want_mean else None, m_host[:B])


def voxelize_deferred(points, voxel_size, pc_range, max_points, max_voxels, want_voxels=False, want_mean=True, algo=None):
    """voxelize_batch without its host read: -> (voxels_cap | None, coords_cap i32[cap,4], num_cap, mean_cap | None,
    m_out i32[B+2] ON THE DEVICE (per-sample counts, total, overflow word), algo used).  The caller reads m_out together
    with whatever other sizes it needs (LidarEncoder.prepare: one read per encoder pass) and slices the cap-sized tensors."""
    _lib.require_gpu(points)
    if points.dtype != torch.float32:
        raise TypeError("points must be float32")
    if points.dim() == 2:
        points = points.unsqueeze(0)
    points = points.contiguous()
    B, N, F = points.shape
    lib = _lib.load()
    cap = lib.ud_voxelize_capacity(B, N, int(max_voxels))
    need = lib.ud_voxelize_workspace_bytes(B, N, int(max_points), int(max_voxels))
    if cap <= 0 or need == 0:
        raise ValueError("invalid voxelization sizes")
    dev = points.device
    ws = _lib.workspace(dev, need, "voxelize")
    voxels = torch.empty((cap, max_points, F), dtype=torch.float32, device=dev) if want_voxels else None
    coords = torch.empty((cap, 4), dtype=torch.int32, device=dev)
    num = torch.empty((cap,), dtype=torch.int32, device=dev)
    mean = torch.empty((cap, F), dtype=torch.float32, device=dev) if want_mean else None
    m_out = torch.empty((B + 2,), dtype=torch.int32, device=dev)
    algo = ALGO if algo is None else algo
    small = _small_algo(ws, B, N, int(max_points), int(max_voxels))
    if algo is None:
        algo = small if small is not None else (1 if B * N < SMALL_CLOUD else 0)
    elif algo == 2:
        if small is None:
            raise ValueError("ud_voxelize: algo 2 takes at most 256 tiles of 1 024 points, P <= 16, B <= 64")
        algo = small
    _forget(ws)
    _lib.check(lib.ud_voxelize(_lib.ptr(points), B, N, F, _f3(voxel_size), _f3(pc_range), int(max_points),
                               int(max_voxels), _lib.ptr(voxels), _lib.ptr(coords), _lib.ptr(num), _lib.ptr(mean),
                               _lib.ptr(m_out), _lib.ptr(ws), ws.numel(), algo, _lib.stream_of(points)), "ud_voxelize")
    # the caller reads m_out[B + 1] and answers with voxelize_confirm (== 0) or voxelize_dirty (refused / overflowed: the workspace is
    # dirty); until then the note is pending and the next small call runs algo 2 (memset first)
    _note_algo(ws, algo, B, N, int(max_points), int(max_voxels), pending=True)
    return voxels, coords, num, mean, m_out, algo


def voxelize_confirm(device):
    """The caller of voxelize_deferred has seen m_out[B + 1] == 0: the pending clean-state notes of this device's voxelizer
    workspaces hold."""
    for ws in _device_workspaces(device):
        pend = getattr(ws, "_ud_clean_pending", None)
        if pend is not None:
            ws._ud_clean, ws._ud_clean_pending = pend, None


def voxelize_dirty(device):
    """A deferred call reported a non-zero overflow word: forget the clean-state notes of THIS device's voxelizer workspaces."""
    for ws in _device_workspaces(device):
        _forget(ws)


class PointToVoxel:
    """Interface of ``spconv.pytorch.utils.PointToVoxel`` as the reference uses it
    (voxelization.py:31-38 ctor kwargs, :54 call): one sample per call, returns
    (voxels f32[M,P,F], coords i32[M,3] as (z,y,x), num_points i32[M])."""

    def __init__(self, vsize_xyz, coors_range_xyz, num_point_features, max_num_voxels,
                 max_num_points_per_voxel, device=None):
        self.vsize = [float(v) for v in vsize_xyz]
        self.range = [float(v) for v in coors_range_xyz]
        self.num_point_features = int(num_point_features)
        self.max_voxels = int(max_num_voxels)
        self.max_points = int(max_num_points_per_voxel)
        self.device = device

    def __call__(self, pc):
        assert pc.dim() == 2 and pc.shape[1] == self.num_point_features
        voxels, coords, num, _, _ = voxelize_batch(pc, self.vsize, self.range, self.max_points,
                                                   self.max_voxels, want_voxels=True, want_mean=False)
        return voxels, coords[:, 1:], num


class Voxelization(nn.Module):
    """Mirror of the reference ``Voxelization`` module (voxelization.py:8-73).

    forward(points) takes a list of per-sample clouds (or one tensor) and returns
    (voxels[M,P,F], voxel_coords i32[M,4] (b,z,y,x), voxel_num_points i32[M]) concatenated over
    the batch.  Equal-length clouds (the collate_fn case) go through ONE batched launch set
    instead of the reference's per-sample python loop + clone + pad + cat.
    ``fused_mean=True`` skips the [M,P,F] tensor and returns the MeanVFE output in its place.
    """

    def __init__(self, voxel_size, point_cloud_range, max_num_points, max_voxels,
                 num_point_features, device=None, fused_mean=False):
        super().__init__()
        assert len(voxel_size) == 3 and len(point_cloud_range) == 6
        self.voxel_size = np.array(voxel_size)
        self.point_cloud_range = np.array(point_cloud_range)
        self.max_num_points = max_num_points
        self.num_point_features = num_point_features
        self._max_voxels = max_voxels
        self.fused_mean = fused_mean
        grid = (self.point_cloud_range[3:6] - self.point_cloud_range[0:3]) / np.array(voxel_size)
        self.grid_size = np.round(grid).astype(np.int64)

    @property
    def max_voxels(self):
        mv = self._max_voxels
        if isinstance(mv, tuple):
            # the reference resolves the tuple in __init__, where self.training is always True
            # (voxelization.py:25-29) -> the training cap is used in both modes
            return mv[0]
        return mv

    def forward(self, points, algo=None):
        """algo: None = the wrapper's choice (fast path, repeated on the hash path after an overflow); 1 = the hash path
        directly (a caller that has already seen the fast path's overflow word)."""
        if not isinstance(points, (list, tuple)):
            points = [points]
        same = all(p.shape == points[0].shape for p in points)
        if same:
            batch = torch.stack(list(points), 0) if len(points) > 1 else points[0].unsqueeze(0)
            vox, coords, num, mean, _ = voxelize_batch(
                batch, self.voxel_size, self.point_cloud_range, self.max_num_points,
                self.max_voxels, want_voxels=not self.fused_mean, want_mean=self.fused_mean, algo=algo)
            return (mean if self.fused_mean else vox), coords, num
        outs = []
        for i, p in enumerate(points):
            vox, coords, num, mean, _ = voxelize_batch(
                p, self.voxel_size, self.point_cloud_range, self.max_num_points, self.max_voxels,
                want_voxels=not self.fused_mean, want_mean=self.fused_mean)
            coords = coords.clone()
            coords[:, 0] = i
            outs.append(((mean if self.fused_mean else vox), coords, num))
        return tuple(torch.cat([o[k] for o in outs], 0) for k in range(3))


class MeanVFE(nn.Module):
    """Mirror of the reference ``MeanVFE`` (mean_vfe.py:6-34).  When the voxelizer ran with
    ``fused_mean`` its input is already the [M,F] mean and is passed through."""

    def __init__(self, num_point_features):
        super().__init__()
        self.num_point_features = num_point_features

    def get_output_feature_dim(self):
        return self.num_point_features

    def forward(self, voxel_features, voxel_num_points, **kwargs):
        if voxel_features.dim() == 2:
            return voxel_features
        s = voxel_features[:, :, :self.num_point_features].sum(dim=1)
        den = torch.clamp_min(voxel_num_points.view(-1, 1), min=1.0).type_as(voxel_features)
        return (s / den).contiguous()
