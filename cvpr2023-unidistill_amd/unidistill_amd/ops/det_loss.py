"""Fused detection-loss terms of the CenterPoint heads (ud_det_focal_*, ud_det_reg_*).

CenterHeadIouAware.get_loss (reference layers/head/det3d/center_head_iou_aware.py:55-298; FocalLoss and
CenterNetRegLoss of layers/losses/det3d.py:287-421) evaluates ~330 small tensor ops per step; here the
focal term over all heat maps and the gathered regression / IoU terms are one forward kernel each, with
the local derivatives kept for a scale-and-scatter backward.
"""
import ctypes

import torch

from .. import _lib

HEADS = ("reg", "height", "dim", "rot", "vel", "iou")       # gathered order: 2+1+3+2+2+1 = 11 values
WIDTH = (2, 1, 3, 2, 2, 1)


def supported(preds, nb):
    """All head tensors fp32 CUDA [B, c, H, W] with contiguous H*W planes, nuScenes code size."""
    if nb != 10 or len(preds) > 8:
        return False
    for pd in preds:
        for name in ("hm",) + HEADS:
            t = pd.get(name)
            if t is None or not t.is_cuda or t.dtype != torch.float32 or t.dim() != 4:
                return False
            if t.stride(3) != 1 or t.stride(2) != t.shape[3] or t.stride(1) != t.shape[2] * t.shape[3]:
                return False
    return True


def _ptr_table(tensors):
    return (ctypes.c_void_p * len(tensors))(*[t.data_ptr() if t is not None else None for t in tensors])


class _FocalFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, gt, alpha, gamma, *hms):
        lib = _lib.load()
        T = len(hms)
        B, _, H, W = hms[0].shape
        ncm = gt.shape[2]
        ncls = [int(h.shape[1]) for h in hms]
        gt = gt.contiguous().float()
        prob = torch.empty((T, B, ncm, H, W), dtype=torch.float32, device=gt.device)
        pos_neg = torch.empty((T, 2), dtype=torch.float32, device=gt.device)
        ws = _lib.workspace(gt.device, lib.ud_det_loss_workspace_bytes(T), "det_loss")
        _lib.check(lib.ud_det_focal_fwd(_ptr_table(hms), (ctypes.c_longlong * T)(*[h.stride(0) for h in hms]),
                                        (ctypes.c_int * T)(*ncls), T, B, ncm, H * W, _lib.ptr(gt), float(alpha),
                                        float(gamma), _lib.ptr(prob), _lib.ptr(pos_neg), _lib.ptr(ws),
                                        ws.numel(), _lib.stream_of(gt)), "ud_det_focal_fwd")
        ctx.save_for_backward(gt, prob)
        ctx.cfg = (ncls, float(alpha), float(gamma))
        return prob, pos_neg[:, 0], pos_neg[:, 1]

    @staticmethod
    def backward(ctx, g_prob, g_pos, g_neg):
        gt, prob = ctx.saved_tensors
        ncls, alpha, gamma = ctx.cfg
        T, B, ncm, H, W = prob.shape
        dl = torch.empty_like(prob)
        gp = None if g_prob is None else g_prob.contiguous().float()
        _lib.check(_lib.load().ud_det_focal_bwd((ctypes.c_int * T)(*ncls), T, B, ncm, H * W, _lib.ptr(gt),
                                                _lib.ptr(prob), _lib.ptr(gp), _lib.ptr(g_pos.contiguous().float()),
                                                _lib.ptr(g_neg.contiguous().float()), alpha, gamma, _lib.ptr(dl),
                                                _lib.stream_of(prob)), "ud_det_focal_bwd")
        return (None, None, None) + tuple(dl[t, :, :ncls[t]] for t in range(T))


def focal_terms(hms, gt, alpha, gamma):
    """hms: T logits tensors [B, ncls_t, H, W]; gt [T, B, ncm, H, W]  ->  (prob [T,B,ncm,H,W], pos [T], neg [T])."""
    return _FocalFn.apply(gt, alpha, gamma, *hms)


class _RegFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, ind, mask, tgt, num_obj, sx, sy, nb, *heads):
        """heads: T * 6 tensors in HEADS order per task."""
        lib = _lib.load()
        T = len(heads) // len(HEADS)
        B, _, H, W = heads[0].shape
        K = ind.shape[2]
        planes, strides = [], []
        for t in range(T):
            for h, wd in zip(heads[t * 6:(t + 1) * 6], WIDTH):
                assert h.shape[1] == wd
                for c in range(wd):
                    planes.append(h[:, c])
                    strides.append(h.stride(0))
        table = (ctypes.c_void_p * len(planes))(*[p.data_ptr() for p in planes])
        ind = ind.contiguous()
        mask_u8 = mask.to(torch.uint8).contiguous()
        tgt = tgt.contiguous().float()
        num_obj = num_obj.contiguous().float()
        dev = ind.device
        losses = torch.empty((T, 12), dtype=torch.float32, device=dev)
        loc = torch.empty((T, B, K, 17), dtype=torch.float32, device=dev)
        ws = _lib.workspace(dev, lib.ud_det_loss_workspace_bytes(T), "det_loss")
        _lib.check(lib.ud_det_reg_fwd(table, (ctypes.c_longlong * len(strides))(*strides), T, B, K, H * W, nb,
                                      _lib.ptr(ind), _lib.ptr(mask_u8), _lib.ptr(tgt), tgt.shape[-1],
                                      _lib.ptr(num_obj), float(sx), float(sy), _lib.ptr(losses), _lib.ptr(loc),
                                      _lib.ptr(ws), ws.numel(), _lib.stream_of(ind)), "ud_det_reg_fwd")
        ctx.save_for_backward(ind, mask_u8, loc)
        ctx.cfg = (T, B, K, H, W, nb)
        return losses[:, :10], losses[:, 10], losses[:, 11]

    @staticmethod
    def backward(ctx, g_box, g_iou, g_aw):
        ind, mask_u8, loc = ctx.saved_tensors
        T, B, K, H, W, nb = ctx.cfg
        dhead = torch.zeros((T, B, 11, H, W), dtype=torch.float32, device=ind.device)
        z = lambda g, shape: (torch.zeros(shape, device=ind.device) if g is None else g.contiguous().float())
        _lib.check(_lib.load().ud_det_reg_bwd(T, B, K, H * W, nb, _lib.ptr(ind), _lib.ptr(mask_u8), _lib.ptr(loc),
                                              _lib.ptr(z(g_box, (T, 10))), _lib.ptr(z(g_iou, (T,))),
                                              _lib.ptr(z(g_aw, (T,))), _lib.ptr(dhead), _lib.stream_of(ind)),
                   "ud_det_reg_bwd")
        grads = []
        for t in range(T):
            c = 0
            for wd in WIDTH:
                grads.append(dhead[t, :, c:c + wd])
                c += wd
        return (None,) * 7 + tuple(grads)


def reg_terms(preds, ind, mask, tgt, num_obj, sx, sy, nb=10):
    """preds: per-task dicts of head tensors; ind/mask [T,B,K], tgt [T,B,K,>=nb], num_obj [T]
    -> (box_loss [T,10], iou_loss [T], iou_aware [T]) normalised like the reference."""
    heads = [pd[name] for pd in preds for name in HEADS]
    return _RegFn.apply(ind, mask, tgt, num_obj, sx, sy, nb, *heads)
