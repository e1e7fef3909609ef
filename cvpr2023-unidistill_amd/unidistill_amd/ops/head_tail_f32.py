"""fp32 mode of the second SepHead convolutions (reference layers/head/det3d/center_head.py:311-362): the 42
(conv3x3 64 -> k <= 3) stacks as ONE grouped convolution on the channels-last fp32 hidden tensor
(ud_head_tail_f32_fwd / _dgrad / _wgrad: exact fp32 FMAs, HBM-bound streaming kernels, deterministic)."""
import os

import torch

from .. import _lib
from . import bn_act, wgrad_stream


def supported(a, head_conv, kmax, k):
    return (a.is_cuda and a.dtype == torch.float32 and a.dim() == 4 and head_conv == 64 and k == 3
            and 1 <= kmax <= 4 and a.shape[1] % 64 == 0)


class _GroupTail(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, weight, bias, G, KM):
        """a [B, G*64, H, W] (any layout; made channels-last), weight [G*KM, 64, 3, 3], bias [G*KM]."""
        _lib.require_gpu(a, weight)
        a = a if a.is_contiguous(memory_format=torch.channels_last) else a.contiguous(memory_format=torch.channels_last)
        B, C, H, W = a.shape
        assert C == G * 64 and weight.shape == (G * KM, 64, 3, 3)
        wt = weight.detach().permute(0, 2, 3, 1).contiguous()          # [G*KM, 3, 3, 64] = [G][KM][9][64]
        z = torch.empty((B, G * KM, H, W), dtype=torch.float32, device=a.device, memory_format=torch.channels_last)
        b = None if bias is None else bias.detach().contiguous()
        _lib.check(_lib.load().ud_head_tail_f32_fwd(_lib.ptr(a), _lib.ptr(wt), _lib.ptr(b), _lib.ptr(z), B, H, W, G,
                                                    KM, _lib.stream_of(a)), "ud_head_tail_f32_fwd")
        ctx.save_for_backward(a, wt)
        ctx.cfg = (G, KM, bias is not None)
        return z

    @staticmethod
    def backward(ctx, dz):
        a, wt = ctx.saved_tensors
        G, KM, has_bias = ctx.cfg
        lib = _lib.load()
        B, C, H, W = a.shape
        dz = dz.float()
        dz = dz if dz.is_contiguous(memory_format=torch.channels_last) else dz.contiguous(memory_format=torch.channels_last)
        da = dw = db = None
        st = _lib.stream_of(a)
        if ctx.needs_input_grad[0]:
            da = torch.empty_like(a)
            _lib.check(lib.ud_head_tail_f32_dgrad(_lib.ptr(dz), _lib.ptr(wt), _lib.ptr(da), B, H, W, G, KM, st),
                       "ud_head_tail_f32_dgrad")
        if ctx.needs_input_grad[1]:
            need = lib.ud_head_tail_f32_wgrad_workspace_bytes(B, H, W, G, KM)
            ws = _lib.workspace(a.device, need, "head_tail_f32")
            dwt = torch.empty((G * KM, 3, 3, 64), dtype=torch.float32, device=a.device)
            _lib.check(lib.ud_head_tail_f32_wgrad(_lib.ptr(a), _lib.ptr(dz), _lib.ptr(dwt), B, H, W, G, KM,
                                                  _lib.ptr(ws), ws.numel(), st), "ud_head_tail_f32_wgrad")
            dw = dwt.permute(0, 3, 1, 2)
        if has_bias and ctx.needs_input_grad[2]:
            db = bn_act.bias_grad(dz)
        return da, dw, db, None, None


def group_tail(a, weight, bias, G, KM):
    return _GroupTail.apply(a, weight, bias, G, KM)


EMIT_COLSUM = True      # False: the first convolution sums its bias gradient itself (ud_colsum_f32 over dy)
FUSED_BN_BWD = True     # False: tail dgrad -> stored gradient -> ud_bn_act_bwd_f32 (the measured alternative)


class _BnReluGroupTail(torch.autograd.Function):
    """relu(bn(y)) -> 42 grouped second convolutions with BatchNorm + ReLU applied as the tail kernels LOAD the first
    convolution's raw output y (ud_head_tail_f32_bn_fwd / _bn_wgrad): the normalised hidden tensor (1.39 GB at B = 4) is
    neither written nor read back -- forward: 2 passes over it instead of 4; backward: the tail's weight gradient reads y, its
    data gradient is recomputed inside the two BatchNorm-backward passes (ud_head_tail_f32_bn_bwd), never stored."""

    @staticmethod
    def forward(ctx, y, gamma, beta, running_mean, running_var, training, momentum, eps, tracked, partial, weight, bias, G, KM):
        _lib.require_gpu(y, weight)
        y = y if y.is_contiguous(memory_format=torch.channels_last) else y.contiguous(memory_format=torch.channels_last)
        B, C, H, W = y.shape
        assert C == G * 64 and weight.shape == (G * KM, 64, 3, 3)
        vec = bn_act.batch_stats(y, gamma, beta, running_mean, running_var, training, momentum, eps, tracked, partial)
        v0, row = vec.data_ptr(), 4 * C
        wt = weight.detach().permute(0, 2, 3, 1).contiguous()          # [G*KM, 3, 3, 64] = [G][KM][9][64]
        z = torch.empty((B, G * KM, H, W), dtype=torch.float32, device=y.device, memory_format=torch.channels_last)
        b = None if bias is None else bias.detach().contiguous()
        _lib.check(_lib.load().ud_head_tail_f32_bn_fwd(_lib.ptr(y), v0 + 3 * row, v0 + 4 * row, _lib.ptr(wt), _lib.ptr(b),
                                                       _lib.ptr(z), B, H, W, G, KM, _lib.stream_of(y)),
                   "ud_head_tail_f32_bn_fwd")
        ctx.save_for_backward(y, vec, wt)
        ctx.cfg = (G, KM, bias is not None, bool(training))
        ctx.weight_param = weight
        return z

    @staticmethod
    def backward(ctx, dz):
        y, vec, wt = ctx.saved_tensors
        G, KM, has_bias, training = ctx.cfg
        if not training:
            raise NotImplementedError("fused BatchNorm backward covers training-mode statistics only")
        lib = _lib.load()
        B, C, H, W = y.shape
        P = B * H * W
        dz = dz.float()
        dz = dz if dz.is_contiguous(memory_format=torch.channels_last) else dz.contiguous(memory_format=torch.channels_last)
        st = _lib.stream_of(y)
        v0, row = vec.data_ptr(), 4 * C
        dy = dgamma = dbeta = dw = db = None
        if ctx.needs_input_grad[10]:
            def wgrad():
                need = lib.ud_head_tail_f32_wgrad_workspace_bytes(B, H, W, G, KM)
                ws = _lib.workspace(y.device, need, "head_tail_f32")
                dwt = torch.empty((G * KM, 3, 3, 64), dtype=torch.float32, device=y.device)
                _lib.check(lib.ud_head_tail_f32_bn_wgrad(_lib.ptr(y), v0 + 3 * row, v0 + 4 * row, _lib.ptr(dz), _lib.ptr(dwt), B, H, W,
                                                         G, KM, _lib.ptr(ws), ws.numel(), _lib.stream_of(y)),
                           "ud_head_tail_f32_bn_wgrad")
                return dwt.permute(0, 3, 1, 2)
            # (reads y and dz only: beside the BatchNorm-backward passes below on the weight-gradient stream, ops/wgrad_stream.py)
            dw = wgrad_stream.defer(ctx.weight_param, wgrad, y, dz, vec) if os.environ.get("UD_TAIL_WGRAD_STREAM", "1") == "1" else wgrad()
        if ctx.needs_input_grad[0] or ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:
            dgb = torch.empty((2, C), dtype=torch.float32, device=y.device)
            g0 = dgb.data_ptr()
            dy = torch.empty_like(y)
            if FUSED_BN_BWD:     # the tail's data gradient is recomputed inside both BatchNorm-backward passes, never stored
                bws = _lib.workspace(y.device, lib.ud_head_tail_f32_bn_bwd_workspace_bytes(B, H, W, G, KM), "head_tail_f32_bn")
                # + the per-channel sums of dy from the pass that stores it: the bias gradient of the first convolution
                # (center_head.py:339: bias=True in front of the BatchNorm) without another pass over the 1.39 GB gradient
                colsum = torch.empty(C, dtype=torch.float32, device=y.device) if EMIT_COLSUM else None
                _lib.check(lib.ud_head_tail_f32_bn_bwd(_lib.ptr(dz), _lib.ptr(wt), y.data_ptr(), v0 + 3 * row, v0 + 4 * row, v0,
                                                       v0 + 2 * row, dy.data_ptr(), g0, g0 + row, _lib.ptr(colsum), B, H, W, G, KM,
                                                       _lib.ptr(bws), bws.numel(), st), "ud_head_tail_f32_bn_bwd")
                if colsum is not None:
                    bn_act.attach_colsum(dy, colsum)
            else:
                da = torch.empty_like(y)                     # gradient at relu(bn(y))
                _lib.check(lib.ud_head_tail_f32_dgrad(_lib.ptr(dz), _lib.ptr(wt), _lib.ptr(da), B, H, W, G, KM, st),
                           "ud_head_tail_f32_dgrad")
                bws = bn_act._workspace(y.device, C)
                _lib.check(lib.ud_bn_act_bwd_f32(y.data_ptr(), None, da.data_ptr(), v0 + 3 * row, v0 + 4 * row, v0,
                                                 v0 + 2 * row, dy.data_ptr(), None, g0, g0 + row, P, C, 1, bws.data_ptr(),
                                                 bws.numel(), st), "ud_bn_act_bwd")
            dgamma, dbeta = dgb[0], dgb[1]
        if has_bias and ctx.needs_input_grad[11]:
            db = bn_act.bias_grad(dz)
        return dy, dgamma, dbeta, None, None, None, None, None, None, None, dw, db, None, None


def bn_relu_group_tail(y, gamma, beta, running_mean, running_var, training, momentum, eps, tracked, partial, weight, bias, G, KM):
    return _BnReluGroupTail.apply(y, gamma, beta, running_mean, running_var, training, momentum, eps, tracked, partial,
                                  weight, bias, G, KM)
