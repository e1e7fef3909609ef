"""The three distillation losses + gaussian box mask, on fused HIP kernels.

Mirrors the module-level functions of the reference's distill experiments (same names, argument
order and meaning), e.g. unidistill/exps/multisensor_fusion/nuscenes/BEVFusion/
BEVFusion_nuscenes_centerhead_camera_exp_distill_lidar.py :196 FeatureDistillLoss,
:248 BEVDistillLoss, :326 ResponseDistillLoss, :100 calculate_box_mask_gaussian,
:73 center_to_corner_box2d (+ the pixel scaling of training_step :478-483).
Gradients flow to the STUDENT tensors only, as in the reference (teacher outputs are detached
because the teacher is frozen).
"""
import ctypes
import os

import torch

from .. import _lib
from ..dist import reduce_mean

_HEAD_ORDER = ("reg", "height", "dim", "rot", "vel", "iou")   # cat order, distill_lidar.py:340-361


def _strides(t):
    return (ctypes.c_int64 * 4)(*t.stride())


def box_corners_bev(gt_boxes, pc_range, voxel_size, out_size_scale):
    """gt_boxes f32[B,M,S] -> (corners f32[B,M,4,2] in BEV pixels, valid bool[B,M]).
    Device version of training_step's host loops (:449-455 valid scan, :466-483 corners)."""
    _lib.require_gpu(gt_boxes)
    gt = gt_boxes.contiguous().float()
    B, M, S = gt.shape
    corners = torch.empty((B, M, 4, 2), dtype=torch.float32, device=gt.device)
    valid = torch.empty((B, M), dtype=torch.uint8, device=gt.device)
    _lib.check(_lib.load().ud_distill_box_corners(
        _lib.ptr(gt), B, M, S, float(pc_range[0]), float(pc_range[1]),
        float(voxel_size[0] * out_size_scale), float(voxel_size[1] * out_size_scale),
        _lib.ptr(corners), _lib.ptr(valid), _lib.stream_of(gt)), "ud_distill_box_corners")
    return corners, valid.bool()


FEATURE_TAP = os.environ.get("UD_FEATURE_TAP", "1") != "0"


class _Tap:
    """Side channel of feature_tap: the box losses leave their (sparse) input gradient here instead of materialising it."""
    __slots__ = ("pending",)

    def __init__(self):
        self.pending = []


class _FeatureTap(torch.autograd.Function):
    """x -> (x for the network, x for the distillation losses), both aliases.  In backward the losses' gradients -- a few
    hundred bilinear footprints per sample -- are ADDED into the dense gradient that came back through the network branch
    (ud_distill_box_bwd_acc) instead of each being written into a zeroed map of the feature's size and summed by autograd
    (4 x 512 x 180 x 180: one 265 MB fill + a three-pass add per feature map and step)."""

    @staticmethod
    def forward(ctx, x, tap):
        ctx.tap = tap
        ctx.set_materialize_grads(False)        # an unused x_loss (plain detector training) costs no zero map and no add
        return x.view_as(x), x.view_as(x)

    @staticmethod
    def backward(ctx, g_net, g_loss):
        tap = ctx.tap
        pending, tap.pending = tap.pending, []
        if g_loss is not None and all(st == 0 for st in g_loss.stride()) and pending:
            g_loss = None                       # _BoxDistill's stride-0 placeholder: its real gradient is in `pending`
        if not pending:
            if g_net is None:
                return g_loss, None
            return (g_net if g_loss is None else g_net + g_loss), None
        g = g_net
        if g is None or g.dtype != torch.float32 or not g.is_cuda or any(st == 0 for st in g.stride()):
            g = torch.zeros_like(pending[0][1], dtype=torch.float32) if g is None else g.float().contiguous()
        elif not _exclusively_owned(g):
            g = g.clone()                       # the producer handed this tensor to another branch too: never add into it
        if g_loss is not None:
            g = g + g_loss                      # a consumer of x_loss other than the box losses
        from . import bn_act
        bn_act.drop_colsum(g)                   # (modified below through its raw pointer: column sums recorded on it are stale)
        lib = _lib.load()
        for kind, sx, tx, corners, valid_u8, gscale in pending:
            B, C, H, W = sx.shape
            M = corners.shape[1]
            ws = _lib.workspace(sx.device, lib.ud_distill_box_bwd_workspace_bytes(B, M, C), "distill_bwd")
            _lib.check(lib.ud_distill_box_bwd_acc(kind, _lib.ptr(sx), _strides(sx), _lib.ptr(tx), _strides(tx), _lib.ptr(corners),
                                                  _lib.ptr(valid_u8), B, M, C, H, W, _lib.ptr(gscale), _lib.ptr(g), _strides(g),
                                                  _lib.ptr(ws), ws.numel(), _lib.stream_of(sx)), "ud_distill_box_bwd_acc")
        return g, None


OWNED_USE_COUNT = 2      # the engine's input vector + this Python wrapper
TAP_STATS = {"in_place": 0, "cloned": 0}


def _exclusively_owned(g):
    """True when nothing but the autograd engine's argument list (and the Python object made for this call) refers to g: a
    producer that returned ONE tensor for two of its inputs (e.g. a residual join handing dy to both branches) leaves a third
    reference in the other branch's input buffer, and an in-place add here would corrupt that branch."""
    count = getattr(g, "_use_count", None)
    ok = count is not None and count() <= OWNED_USE_COUNT and g._base is None
    TAP_STATS["in_place" if ok else "cloned"] += 1
    return ok


def feature_tap(x):
    """(x_net, x_loss): feed x_net on into the network and x_loss to FeatureDistillLoss / BEVDistillLoss (fp32 CUDA training only;
    anything else gets (x, x) and the losses their usual dense gradient)."""
    if not (FEATURE_TAP and torch.is_grad_enabled() and x.requires_grad and x.is_cuda and x.dtype == torch.float32):
        return x, x
    tap = _Tap()
    x_net, x_loss = _FeatureTap.apply(x, tap)
    x_loss._ud_tap = tap
    return x_net, x_loss


class _BoxDistill(torch.autograd.Function):
    @staticmethod
    def forward(ctx, kind, student, teacher, corners, valid_u8, weight, tap=None):
        _lib.require_gpu(student, teacher, corners, valid_u8)
        s = student if student.dtype == torch.float32 else student.float()
        t = teacher.detach()
        t = t if t.dtype == torch.float32 else t.float()
        B, C, H, W = s.shape
        M = corners.shape[1]
        box_loss = torch.empty((B, M), dtype=torch.float32, device=s.device)
        _lib.check(_lib.load().ud_distill_box_fwd(kind, _lib.ptr(s), _strides(s), _lib.ptr(t),
                                                  _strides(t), _lib.ptr(corners), _lib.ptr(valid_u8),
                                                  B, M, C, H, W, _lib.ptr(box_loss),
                                                  _lib.stream_of(s)), "ud_distill_box_fwd")
        den = weight + 1e-4
        ctx.save_for_backward(s, t, corners, valid_u8, den)
        ctx.kind = kind
        ctx.tap = tap if s is student else None
        return box_loss.sum() / den

    @staticmethod
    def backward(ctx, gloss):
        s, t, corners, valid_u8, den = ctx.saved_tensors
        B, C, H, W = s.shape
        M = corners.shape[1]
        gscale = (gloss / den).reshape(1).float().contiguous()
        if ctx.tap is not None:
            # the tap adds this gradient into the network branch's map; what autograd gets here is a stride-0 zero of the right shape
            ctx.tap.pending.append((ctx.kind, s, t, corners, valid_u8, gscale))
            return None, torch.zeros((), dtype=s.dtype, device=s.device).expand_as(s), None, None, None, None, None
        gs = torch.zeros_like(s)
        lib = _lib.load()
        ws = _lib.workspace(s.device, lib.ud_distill_box_bwd_workspace_bytes(B, M, C), "distill_bwd")
        _lib.check(lib.ud_distill_box_bwd(ctx.kind, _lib.ptr(s), _strides(s), _lib.ptr(t),
                                          _strides(t), _lib.ptr(corners), _lib.ptr(valid_u8),
                                          B, M, C, H, W, _lib.ptr(gscale), _lib.ptr(gs),
                                          _strides(gs), _lib.ptr(ws), ws.numel(), _lib.stream_of(s)),
                   "ud_distill_box_bwd")
        return None, gs, None, None, None, None, None


def _box_loss(kind, student, teacher, coords, indices, weight):
    corners = coords.contiguous().float()
    valid_u8 = indices.to(torch.uint8).contiguous()
    if weight is None:
        weight = reduce_mean(indices.float().sum())
    return _BoxDistill.apply(kind, student, teacher, corners, valid_u8, weight, getattr(student, "_ud_tap", None))


def FeatureDistillLoss(feature_lidar, feature_fuse, gt_boxes_bev_coords, gt_boxes_indices, weight=None):
    """student map, teacher map [B,C,H,W]; coords f32[B,M,4,2] (BEV px); indices bool[B,M]."""
    return _box_loss(0, feature_lidar, feature_fuse, gt_boxes_bev_coords, gt_boxes_indices, weight)


def BEVDistillLoss(bev_lidar, bev_fuse, gt_boxes_bev_coords, gt_boxes_indices, weight=None):
    return _box_loss(1, bev_lidar, bev_fuse, gt_boxes_bev_coords, gt_boxes_indices, weight)


def calculate_box_mask_gaussian(preds_shape, target, pc_range, voxel_size, out_size_scale):
    """target: gt boxes f32[B,M,S] ON THE DEVICE (the reference takes a host numpy copy)."""
    _lib.require_gpu(target)
    gt = target.contiguous().float()
    B, M, S = gt.shape
    H, W = int(preds_shape[2]), int(preds_shape[3])
    lib = _lib.load()
    mask = torch.empty((B, H, W), dtype=torch.float32, device=gt.device)
    ws = _lib.workspace(gt.device, lib.ud_distill_mask_workspace_bytes(B, M), "distill_mask")
    _lib.check(lib.ud_distill_gaussian_mask(
        _lib.ptr(gt), B, M, S, float(pc_range[0]), float(pc_range[1]),
        float(voxel_size[0] * out_size_scale), float(voxel_size[1] * out_size_scale), H, W,
        _lib.ptr(mask), _lib.ptr(ws), ws.numel(), _lib.stream_of(gt)), "ud_distill_gaussian_mask")
    return mask


def _ptr_array(tensors):
    return (ctypes.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])


def _in_place(t):
    """fp32 CUDA map [B, ch, H, W] whose pixels are linear in memory (stride(H) == W * stride(W)): dense NCHW, channels-last, or a
    channel slice of a channels-last tensor -- read by the kernels through (sb, sc, sp) strides; anything else gets a dense copy."""
    if t.dtype != torch.float32 or not (t.stride(2) == t.shape[3] * t.stride(3) or t.shape[2] == 1):
        t = t.contiguous().float()
    return t


def _stride_array(tensors):
    flat = []
    for t in tensors:
        flat += [t.stride(0), t.stride(1), t.stride(3)]
    return (ctypes.c_int64 * len(flat))(*flat)


class _RespDistill(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mask, weight, clamp, n_hm, *tensors):
        # tensors = student hm[n_hm], student reg[n_reg], teacher hm[n_hm], teacher reg[n_reg]; all read in place (the head outputs
        # are channel slices of one packed channels-last tensor: 84 dense copies per step otherwise)
        n_reg = len(tensors) // 2 - n_hm
        s_hm = [_in_place(t) for t in tensors[:n_hm]]
        s_reg = [_in_place(t) for t in tensors[n_hm:n_hm + n_reg]]
        t_hm = [_in_place(t.detach()) for t in tensors[n_hm + n_reg:2 * n_hm + n_reg]]
        t_reg = [_in_place(t.detach()) for t in tensors[2 * n_hm + n_reg:]]
        B, _, H, W = s_hm[0].shape
        hm_ch = (ctypes.c_int * n_hm)(*[t.shape[1] for t in s_hm])
        reg_ch = (ctypes.c_int * n_reg)(*[t.shape[1] for t in s_reg])
        nblk = B * ((H * W + 255) // 256)
        partial = torch.empty((nblk, 2), dtype=torch.float32, device=mask.device)
        _lib.check(_lib.load().ud_distill_resp_fwd_strided(
            _ptr_array(s_hm), _stride_array(s_hm), _ptr_array(t_hm), _stride_array(t_hm), hm_ch, n_hm,
            _ptr_array(s_reg), _stride_array(s_reg), _ptr_array(t_reg), _stride_array(t_reg), reg_ch, n_reg,
            _lib.ptr(mask), B, H, W, float(clamp), float(1.0 - clamp), _lib.ptr(partial), _lib.stream_of(mask)),
            "ud_distill_resp_fwd_strided")
        den = weight + 1e-4
        sums = partial.sum(0)
        ctx.save_for_backward(mask, den, *s_hm, *s_reg, *t_hm, *t_reg)
        ctx.cfg = (clamp, n_hm, n_reg)
        return sums[0] / den, sums[1] / den

    @staticmethod
    def backward(ctx, g_cls, g_reg):
        clamp, n_hm, n_reg = ctx.cfg
        mask, den = ctx.saved_tensors[:2]
        rest = ctx.saved_tensors[2:]
        s_hm, s_reg = rest[:n_hm], rest[n_hm:n_hm + n_reg]
        t_hm, t_reg = rest[n_hm + n_reg:2 * n_hm + n_reg], rest[2 * n_hm + n_reg:]
        B, _, H, W = s_hm[0].shape
        # all student-side gradients are carved from ONE allocation, each a dense NCHW map (a thread per pixel writes coalesced
        # rows; channel slices of a channels-last buffer made this kernel 0.42 ms of 4-byte stores at 168-byte stride)
        chs = [t.shape[1] for t in s_hm] + [t.shape[1] for t in s_reg]
        flat = torch.empty((B * H * W * sum(chs),), dtype=torch.float32, device=mask.device)
        views, o = [], 0
        for ch in chs:
            views.append(flat[o:o + B * ch * H * W].view(B, ch, H, W))
            o += B * ch * H * W
        gh, gr = views[:n_hm], views[n_hm:]
        sc = (g_cls / den).reshape(1).float().contiguous()
        sr = (g_reg / den).reshape(1).float().contiguous()
        hm_ch = (ctypes.c_int * n_hm)(*[t.shape[1] for t in s_hm])
        reg_ch = (ctypes.c_int * n_reg)(*[t.shape[1] for t in s_reg])
        _lib.check(_lib.load().ud_distill_resp_bwd_strided(
            _ptr_array(s_hm), _stride_array(s_hm), _ptr_array(t_hm), _stride_array(t_hm), _ptr_array(gh), _stride_array(gh),
            hm_ch, n_hm, _ptr_array(s_reg), _stride_array(s_reg), _ptr_array(t_reg), _stride_array(t_reg), _ptr_array(gr),
            _stride_array(gr), reg_ch, n_reg, _lib.ptr(mask), B, H, W, float(clamp), float(1.0 - clamp), _lib.ptr(sc),
            _lib.ptr(sr), _lib.stream_of(mask)), "ud_distill_resp_bwd_strided")
        return (None, None, None, None, *gh, *gr, *([None] * (n_hm + n_reg)))


def ResponseDistillLoss(resp_lidar, resp_fuse, gt_boxes, pc_range, voxel_size, out_size_scale,
                        clamp=1e-4, weight=None, mask=None):
    """resp_lidar / resp_fuse: per-task dicts with keys hm, reg, height, dim, rot, vel, iou.
    Student ``hm`` is the head's clamped-sigmoid output, teacher ``hm`` is a logit.
    Returns (loss_cls_distill, loss_reg_distill).  ``clamp`` is the _sigmoid clamp of the calling
    experiment (1e-4, or 1e-3 in camera_exp_distill_fusion.py:191)."""
    s_hm = [d["hm"] for d in resp_lidar]
    t_hm = [d["hm"] for d in resp_fuse]
    s_reg = [d[k] for d in resp_lidar for k in _HEAD_ORDER]
    t_reg = [d[k] for d in resp_fuse for k in _HEAD_ORDER]
    if mask is None:
        mask = calculate_box_mask_gaussian(s_reg[0].shape, gt_boxes, pc_range, voxel_size, out_size_scale)
    if weight is None:
        weight = reduce_mean(mask.sum())
    return _RespDistill.apply(mask, weight, clamp, len(s_hm), *s_hm, *s_reg, *t_hm, *t_reg)
