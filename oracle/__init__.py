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
