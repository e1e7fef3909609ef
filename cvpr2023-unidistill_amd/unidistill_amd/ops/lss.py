"""Camera lift-splat ops: frustum geometry + binning, depth softmax, lift, fused lift+splat.

Reference: unidistill/layers/blocks_3d/mmdet3d/lss_fpn.py:173-240 (frustum/geometry), :289-316
(softmax (x) context, permute, binning, voxel_pooling).
"""
import ctypes

import numpy as np
import torch

from .. import _lib
from . import bev_pool as _bp


def _f3(v):
    return (ctypes.c_float * 3)(*[float(x) for x in v])


def bin_origin_fp32(voxel_coord, voxel_size):
    """(voxel_coord - voxel_size / 2) evaluated in fp32 exactly as lss_fpn.py:311-312 does."""
    vc = np.asarray(voxel_coord, np.float32)
    vs = np.asarray(voxel_size, np.float32)
    return (vc - vs / np.float32(2.0)).astype(np.float32), vs


def prepare_mats(sensor2ego, intrin, ida, bda, ida_inv=None, intrin_inv=None):
    """[B,ncam,4,4] x3 (+ bda [B,4,4] or None) -> mats f32[B*ncam,3,16] on the GPU.

    ida_inv / intrin_inv: the fp32 inverses exactly as the reference's ``ida_mat.inverse()`` /
    ``torch.inverse(intrin_mat)`` (lss_fpn.py:222,233) produced them; with them the ego coordinates
    and bins are bit-identical to the reference's CPU output.  None -> correctly rounded inverse
    (fp64 Gauss-Jordan in the kernel; no solver launch, no host sync)."""
    _lib.require_gpu(sensor2ego, intrin, ida, bda, ida_inv, intrin_inv)
    B, ncam = sensor2ego.shape[:2]
    s2e, k, a = (t.contiguous().float() for t in (sensor2ego, intrin, ida))
    bd, ai, ki = (None if t is None else t.contiguous().float() for t in (bda, ida_inv, intrin_inv))
    mats = torch.empty((B * ncam, 3, 16), dtype=torch.float32, device=s2e.device)
    _lib.check(_lib.load().ud_lss_prepare_mats(_lib.ptr(s2e), _lib.ptr(k), _lib.ptr(a), _lib.ptr(bd),
                                               _lib.ptr(ai), _lib.ptr(ki), B, ncam, _lib.ptr(mats),
                                               _lib.stream_of(s2e)), "ud_lss_prepare_mats")
    return mats


def geometry(mats, fu, fv, fd, B, ncam, lo, size, has_bda=True, want_geom=False):
    """-> (bins i32[B, ncam*D*fH*fW, 3], geom f32[B,ncam,D,fH,fW,3] | None)."""
    D, fH, fW = fd.numel(), fv.numel(), fu.numel()
    dev = mats.device
    bins = torch.empty((B, ncam * D * fH * fW, 3), dtype=torch.int32, device=dev)
    geom = torch.empty((B, ncam, D, fH, fW, 3), dtype=torch.float32, device=dev) if want_geom else None
    _lib.check(_lib.load().ud_lss_geometry(_lib.ptr(mats), _lib.ptr(fu), _lib.ptr(fv), _lib.ptr(fd),
                                           B, ncam, D, fH, fW, _f3(lo), _f3(size), 1 if has_bda else 0,
                                           _lib.ptr(geom), _lib.ptr(bins), _lib.stream_of(mats)),
               "ud_lss_geometry")
    return bins, geom


def depth_ctx(depth_feature, D, C):
    """depth_feature f32[BN, D+C, fH, fW] (any strides) -> prob [BN,D,fH*fW], ctx_pm [BN,fH*fW,C]."""
    _lib.require_gpu(depth_feature)
    BN, ch, fH, fW = depth_feature.shape
    assert ch >= D + C
    x = depth_feature if depth_feature.dtype == torch.float32 else depth_feature.float()
    sn, sc, sh, sw = x.stride()
    if sw * fW != sh:
        x = x.contiguous()
        sn, sc, sh, sw = x.stride()
    prob = torch.empty((BN, D, fH * fW), dtype=torch.float32, device=x.device)
    ctx = torch.empty((BN, fH * fW, C), dtype=torch.float32, device=x.device)
    _lib.check(_lib.load().ud_lss_depth_ctx(_lib.ptr(x), sn, sc, sh, sw, BN, D, C, fH, fW,
                                            _lib.ptr(prob), _lib.ptr(ctx), _lib.stream_of(x)),
               "ud_lss_depth_ctx")
    return prob, ctx


def _dense_permutation(shape, strides):
    """True when (shape, strides) is a non-overlapping dense layout (some permutation of contiguous)."""
    expect = 1
    for n, st in sorted(((n, st) for n, st in zip(shape, strides) if n > 1), key=lambda t: t[1]):
        if st != expect:
            return False
        expect *= n
    return True


def _lift_bwd(gsrc, pos, prob, ctx, like, ncam, D, C, nx, ny):
    """like = (shape, strides, device) of the forward's depth_feature (only its meta data is kept, the
    activation itself is not held alive).  Channels beyond D+C are never written by the kernel and get
    a zero gradient whichever layout branch allocates the buffer."""
    shape, strides, device = like
    BN, ch, fH, fW = shape
    alloc = torch.zeros if ch > D + C else torch.empty
    if strides[3] * fW == strides[2] and _dense_permutation(shape, strides):
        g = alloc(BN * ch * fH * fW, dtype=torch.float32, device=device).as_strided(shape, strides)
    else:
        g = alloc(shape, dtype=torch.float32, device=device)
    sn, sc, sh, sw = g.stride()
    lib = _lib.load()
    need = lib.ud_lss_lift_bwd_workspace_bytes(BN, D, C, fH, fW)
    ws = _lib.workspace(device, need, "lss_bwd")
    _lib.check(lib.ud_lss_lift_bwd(_lib.ptr(gsrc), _lib.ptr(pos), _lib.ptr(prob), _lib.ptr(ctx),
                                   _lib.ptr(g), sn, sc, sh, sw, BN, ncam, D, C, fH, fW, nx, ny,
                                   _lib.ptr(ws), ws.numel(), _lib.stream_of(g)), "ud_lss_lift_bwd")
    return g


class Lift(torch.autograd.Function):
    """Materialised lift (reference boundary): depth_feature [BN,D+C,fH,fW] ->
    [BN, D, fH, fW, C] == (softmax(depth) (x) context).permute(...,2 last)  (lss_fpn.py:289-310)."""

    @staticmethod
    def forward(ctx, depth_feature, D, C):
        prob, cpm = depth_ctx(depth_feature, D, C)
        BN, _, fH, fW = depth_feature.shape
        lifted = torch.empty((BN, D, fH, fW, C), dtype=torch.float32, device=depth_feature.device)
        _lib.check(_lib.load().ud_lss_lift_fwd(_lib.ptr(prob), _lib.ptr(cpm), _lib.ptr(lifted), BN, D,
                                               C, fH, fW, _lib.stream_of(lifted)), "ud_lss_lift_fwd")
        ctx.save_for_backward(prob, cpm)
        ctx.meta = ((tuple(depth_feature.shape), tuple(depth_feature.stride()), depth_feature.device), D, C)
        ctx.mark_non_differentiable()
        return lifted

    @staticmethod
    def backward(ctx, g_lifted):
        prob, cpm = ctx.saved_tensors
        like, D, C = ctx.meta
        g = _lift_bwd(g_lifted.contiguous().float(), None, prob, cpm, like, 1, D, C, 1, 1)
        return g, None, None


class LiftSplat(torch.autograd.Function):
    """Fused lift + splat: depth_feature [B*ncam, D+C, fH, fW] + bins i32[B,N,3] -> BEV map, a
    [B, C, ny, nx] view of an NHWC buffer (same return convention as VoxelPooling).  The
    [B,N,C] tensor (484 MB at the BASELINE shape) is never materialised, forward or backward."""

    @staticmethod
    def forward(ctx, depth_feature, bins, B, ncam, D, C, nx, ny, nz):
        """bins: i32[B, N, 3], or a frustum description (mats, fu, fv, fd, lo, size, has_bda) as ops.lss.geometry takes it --
        the bins are then computed inside the list-building kernel and never written (ud_lss_splat_geom_fwd)."""
        frustum = bins if isinstance(bins, tuple) else None
        _lib.require_gpu(depth_feature, *(frustum[:4] if frustum else (bins,)))
        prob, cpm = depth_ctx(depth_feature, D, C)
        BN, _, fH, fW = depth_feature.shape
        N = ncam * D * fH * fW
        assert BN == B * ncam
        if frustum is None:
            assert bins.dtype == torch.int32 and bins.is_contiguous() and bins.shape[0] == B and bins.shape[1] == N
        dev = depth_feature.device
        out = torch.empty((B, ny, nx, C), dtype=torch.float32, device=dev)
        pos = torch.empty((B, N, 3), dtype=torch.int32, device=dev)
        lib = _lib.load()
        need = lib.ud_bev_pool_workspace_bytes(B, N, C, nx, ny, nz)
        ws = _lib.workspace(dev, need, "bev_pool")
        if frustum is not None:
            mats, fu, fv, fd, lo, size, has_bda = frustum
            assert fd.numel() == D and fv.numel() == fH and fu.numel() == fW and mats.shape[0] == BN
            _lib.check(lib.ud_lss_splat_geom_fwd(_lib.ptr(mats), _lib.ptr(fu), _lib.ptr(fv), _lib.ptr(fd), _f3(lo), _f3(size),
                                                 1 if has_bda else 0, _lib.ptr(prob), _lib.ptr(cpm), _lib.ptr(out),
                                                 _lib.ptr(pos), B, ncam, D, fH, fW, C, nx, ny, nz,
                                                 _lib.ptr(ws), ws.numel(), _lib.stream_of(out)),
                       "ud_lss_splat_geom_fwd")
        else:
            _lib.check(lib.ud_lss_splat_fwd(_lib.ptr(bins), _lib.ptr(prob), _lib.ptr(cpm), _lib.ptr(out),
                                            _lib.ptr(pos), B, ncam, D, fH, fW, C, nx, ny, nz,
                                            _lib.ptr(ws), ws.numel(), _lib.stream_of(out)),
                       "ud_lss_splat_fwd")
        ctx.save_for_backward(prob, cpm, pos)
        ctx.meta = ((tuple(depth_feature.shape), tuple(depth_feature.stride()), depth_feature.device),
                    ncam, D, C, nx, ny)
        if frustum is None:
            ctx.mark_non_differentiable(bins)
        return out.permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, gout):
        prob, cpm, pos = ctx.saved_tensors
        like, ncam, D, C, nx, ny = ctx.meta
        g_nhwc = gout.permute(0, 2, 3, 1)
        if not g_nhwc.is_contiguous() or g_nhwc.dtype != torch.float32:
            g_nhwc = g_nhwc.contiguous().float()
        g = _lift_bwd(g_nhwc, pos, prob, cpm, like, ncam, D, C, nx, ny)
        return g, None, None, None, None, None, None, None, None


lift = Lift.apply
lift_splat = LiftSplat.apply
