"""BEV pool: the reference's ``voxel_pooling_ext`` boundary and its ``VoxelPooling`` Function.

Reference: unidistill/layers/blocks_3d/mmdet3d/lss_fpn.py:12-82.
"""
import torch

from .. import _lib

POOL_ACCUMULATE = 0
POOL_OVERWRITE = 1


def voxel_pooling_forward_wrapper(batch_size, num_points, num_channels, nx, ny, nz,
                                  geom_xyz, input_features, output_features, pos_memo):
    """Same signature as the reference's missing pybind extension (lss_fpn.py:48-59).

    ``output_features`` f32[B,ny,nx,C] must be pre-zeroed and ``pos_memo`` i32[B,N,3] pre-set to
    -1 by the caller exactly as the reference wrapper does; both are written in place.
    """
    _pool_fwd(geom_xyz, input_features, output_features, pos_memo, int(batch_size),
              int(num_points), int(num_channels), int(nx), int(ny), int(nz), POOL_ACCUMULATE)
    return 0


def _pool_fwd(geom, feat, out, pos, B, N, C, nx, ny, nz, flags):
    _lib.require_gpu(geom, feat, out, pos)
    if geom.dtype != torch.int32 or pos.dtype != torch.int32:
        raise TypeError("geom_xyz / pos_memo must be int32")
    if feat.dtype != torch.float32 or out.dtype != torch.float32:
        raise TypeError("features must be float32")
    for t in (geom, feat, out, pos):
        if not t.is_contiguous():
            raise ValueError("voxel pooling tensors must be contiguous (lss_fpn.py:30-31)")
    lib = _lib.load()
    need = lib.ud_bev_pool_workspace_bytes(B, N, C, nx, ny, nz)
    if need == 0:
        raise ValueError("invalid voxel pooling sizes")
    ws = _lib.workspace(feat.device, need, "bev_pool")
    _lib.check(lib.ud_bev_pool_fwd(_lib.ptr(geom), _lib.ptr(feat), _lib.ptr(out), _lib.ptr(pos),
                                   B, N, C, nx, ny, nz, flags, _lib.ptr(ws), ws.numel(),
                                   _lib.stream_of(feat)), "ud_bev_pool_fwd")


def _pool_bwd(gout, pos, B, N, C, nx, ny):
    """gout: logical [B, C, ny, nx] with any strides -> gfeat f32[B, N, C]."""
    _lib.require_gpu(gout, pos)
    lib = _lib.load()
    if gout.dtype != torch.float32:
        gout = gout.float()
    gfeat = torch.empty((B, N, C), dtype=torch.float32, device=gout.device)
    sb, sc, sy, sx = gout.stride()
    need = lib.ud_bev_pool_bwd_workspace_bytes(B, C, nx, ny, sc)
    ws = _lib.workspace(gout.device, need, "bev_pool_bwd")
    _lib.check(lib.ud_bev_pool_bwd(_lib.ptr(gout), sb, sc, sy, sx, _lib.ptr(pos), _lib.ptr(gfeat),
                                   B, N, C, nx, ny, _lib.ptr(ws), ws.numel(),
                                   _lib.stream_of(gout)), "ud_bev_pool_bwd")
    return gfeat


class VoxelPooling(torch.autograd.Function):
    """Drop-in for the reference ``VoxelPooling`` (lss_fpn.py:12-79).

    forward(geom_xyz i32[B,...,3], input_features f32[B,...,C], voxel_num[3]) -> f32 view
    [B, C, ny, nx] (a permute of the NHWC result, like the reference).  The 484 MB zero gradient
    buffer the reference allocates in forward (lss_fpn.py:34) is not needed and not allocated.
    """

    @staticmethod
    def forward(ctx, geom_xyz, input_features, voxel_num):
        assert geom_xyz.is_contiguous()
        assert input_features.is_contiguous()
        ctx.mark_non_differentiable(geom_xyz)
        B = input_features.shape[0]
        C = input_features.shape[-1]
        geom = geom_xyz.reshape(B, -1, 3)
        feat = input_features.reshape(B, -1, C)
        assert geom.shape[1] == feat.shape[1]
        N = feat.shape[1]
        nx, ny, nz = (int(v) for v in voxel_num)
        out = torch.empty((B, ny, nx, C), dtype=feat.dtype, device=feat.device)
        pos = torch.empty((B, N, 3), dtype=torch.int32, device=feat.device)
        _pool_fwd(geom, feat, out, pos, B, N, C, nx, ny, nz, POOL_OVERWRITE)
        ctx.save_for_backward(pos)
        ctx.meta = (tuple(input_features.shape), B, N, C, nx, ny)
        return out.permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, grad_output_features):
        (pos,) = ctx.saved_tensors
        shape, B, N, C, nx, ny = ctx.meta
        gfeat = _pool_bwd(grad_output_features, pos, B, N, C, nx, ny)
        return None, gfeat.reshape(shape), None


voxel_pooling = VoxelPooling.apply
