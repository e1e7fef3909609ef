"""CPU oracle for the UniDistill hot path -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this package (see ``ud_oracle.c`` header).  It wraps ``oracle/_build/libud_oracle.so`` (plain C,
built by ``oracle/Makefile``) behind numpy arrays and adds a few numpy-only restatements.
"""
import ctypes
import os
import subprocess

import numpy as np

_DIR = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_DIR, "_build", "libud_oracle.so")
_lib = None


def build(force=False):
    """Compile ud_oracle.c with gcc (idempotent)."""
    src = os.path.join(_DIR, "ud_oracle.c")
    stale = (not os.path.exists(_SO)) or os.path.getmtime(_SO) < os.path.getmtime(src)
    if force or stale:
        subprocess.check_call(["make", "-s", "-C", _DIR])
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_SO)
    return _lib


def _p(a, ct):
    return a.ctypes.data_as(ctypes.POINTER(ct))


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


F = ctypes.c_float
I = ctypes.c_int32


def bev_pool_fwd(geom, feat, nx, ny, nz):
    """geom i32[B,N,3], feat f32[B,N,C] -> (out f32[B,ny,nx,C], pos i32[B,N,3])."""
    geom, feat = _i32(geom), _f32(feat)
    B, N, C = feat.shape
    out = np.zeros((B, ny, nx, C), np.float32)
    pos = np.full((B, N, 3), -1, np.int32)
    lib().oracle_bev_pool_fwd(_p(geom, I), _p(feat, F), _p(out, F), _p(pos, I), B, N, C, nx, ny, nz)
    return out, pos


def bev_pool_bwd(gout_nchw, pos):
    """gout f32[B,C,ny,nx], pos i32[B,N,3] -> gfeat f32[B,N,C]."""
    gout_nchw, pos = _f32(gout_nchw), _i32(pos)
    B, C, ny, nx = gout_nchw.shape
    N = pos.shape[1]
    gfeat = np.empty((B, N, C), np.float32)
    lib().oracle_bev_pool_bwd(_p(gout_nchw, F), _p(pos, I), _p(gfeat, F), B, N, C, nx, ny)
    return gfeat


def voxelize(points, voxel_size, pc_range, max_points, max_voxels, with_voxels=True):
    """points f32[B,N,F] -> dict(voxels[M,P,F]|None, coords i32[M,4] (b,z,y,x), num i32[M],
    mean f32[M,F], m i32[B+1])."""
    points = _f32(points)
    if points.ndim == 2:
        points = points[None]
    B, N, Fd = points.shape
    cap = min(B * max_voxels, B * N)
    vs, rg = _f32(voxel_size), _f32(pc_range)
    voxels = np.zeros((cap, max_points, Fd), np.float32) if with_voxels else None
    coords = np.zeros((cap, 4), np.int32)
    num = np.zeros((cap,), np.int32)
    mean = np.zeros((cap, Fd), np.float32)
    m = np.zeros((B + 1,), np.int32)
    fn = lib().oracle_voxelize
    fn.restype = ctypes.c_int
    M = fn(_p(points, F), B, N, Fd, _p(vs, F), _p(rg, F), max_points, max_voxels,
           _p(voxels, F) if with_voxels else None, _p(coords, I), _p(num, I), _p(mean, F), _p(m, I))
    return dict(voxels=voxels[:M] if with_voxels else None, coords=coords[:M], num=num[:M],
                mean=mean[:M], m=m)


def mean_vfe(voxels, num):
    voxels, num = _f32(voxels), _i32(num)
    M, P, Fd = voxels.shape
    out = np.empty((M, Fd), np.float32)
    lib().oracle_mean_vfe(_p(voxels, F), _p(num, I), _p(out, F), M, P, Fd)
    return out
