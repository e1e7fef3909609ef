"""BatchNorm2d (+ residual) (+ ReLU) on channels-last bf16 activations (ud_bn_act_*, ud_head_tail_stats).

The BatchNorm -> ReLU links of the reference's dense layers (base_bev_backbone.py:48-66,
center_head.py:408-420, mmdet ResNet bottlenecks) as one statistics pass + one streaming pass; the
backward recomputes the ReLU mask from x instead of storing it.
"""
import torch

from .. import _lib


def supported(x, bn):
    """4-D channels-last maps [B, C, H, W] or 2-D row tensors [M, C] (sparse voxel features), bf16."""
    if not (x.is_cuda and x.dtype == torch.bfloat16 and x.dim() in (2, 4) and x.shape[1] % 16 == 0
            and x.shape[0] > 0 and bn.affine and bn.track_running_stats and bn.momentum is not None):
        return False
    return x.is_contiguous() if x.dim() == 2 else x.is_contiguous(memory_format=torch.channels_last)


def _like(t, ref):
    """t in ref's dtype/layout (bf16; contiguous rows or channels-last map)."""
    t = t.to(torch.bfloat16)
    return t.contiguous() if ref.dim() == 2 else t.contiguous(memory_format=torch.channels_last)


class _BnActFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, residual, running_mean, running_var, training, momentum, eps, relu,
                tracked=None):
        lib = _lib.load()
        _lib.require_gpu(x, gamma, beta)
        C = x.shape[1]
        P = x.numel() // C
        dev = x.device
        g32 = gamma.detach() if gamma.dtype == torch.float32 else gamma.detach().float()
        b32 = beta.detach() if beta.dtype == torch.float32 else beta.detach().float()
        stream = _lib.stream_of(x)
        if training:
            vec = torch.empty((5, C), dtype=torch.float32, device=dev)
            mean, var, invstd, scale, shift = vec[0], vec[1], vec[2], vec[3], vec[4]
            ws = _lib.workspace(dev, lib.ud_bn_act_workspace_bytes(C), "bn_act")
            fp32_buffers = running_mean is not None and running_mean.dtype == torch.float32
            rm, rv = (running_mean, running_var) if fp32_buffers else (None, None)   # updated in-kernel
            _lib.check(lib.ud_bn_stats(_lib.ptr(x), P, C, _lib.ptr(g32), _lib.ptr(b32), float(eps),
                                       _lib.ptr(mean), _lib.ptr(var), _lib.ptr(invstd), _lib.ptr(scale),
                                       _lib.ptr(shift), _lib.ptr(rm), _lib.ptr(rv), float(momentum or 0.0),
                                       _lib.ptr(tracked), _lib.ptr(ws), ws.numel(), stream), "ud_bn_stats")
            if running_mean is not None and not fp32_buffers:
                with torch.no_grad():
                    running_mean.mul_(1 - momentum).add_(mean, alpha=momentum)
                    running_var.mul_(1 - momentum).add_(var, alpha=momentum * P / max(P - 1, 1))
        else:
            invstd = torch.rsqrt(running_var.float() + eps)
            mean = running_mean.float()
            scale = (g32 * invstd).contiguous()
            shift = (b32 - mean * scale).contiguous()
        if residual is not None:
            residual = _like(residual, x)
        y = torch.empty_like(x)
        _lib.check(lib.ud_bn_act_fwd(_lib.ptr(x), _lib.ptr(residual), _lib.ptr(scale), _lib.ptr(shift),
                                     _lib.ptr(y), P, C, 1 if relu else 0, stream), "ud_bn_act_fwd")
        ctx.cfg = (bool(training), bool(relu), residual is not None)
        ctx.save_for_backward(x, y if (residual is not None and relu) else None, scale, shift, mean, invstd)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y, scale, shift, mean, invstd = ctx.saved_tensors
        training, relu, has_res = ctx.cfg
        if not training:
            raise NotImplementedError("fused BatchNorm backward covers training-mode statistics only")
        lib = _lib.load()
        C = x.shape[1]
        P = x.numel() // C
        dy = _like(dy, x)
        dx = torch.empty_like(x)
        dres = torch.empty_like(x) if (has_res and ctx.needs_input_grad[3]) else None
        if dres is not None and not relu:
            dres = None                                   # no mask: the residual gradient is dy itself
        dgb = torch.empty((2, C), dtype=torch.float32, device=x.device)
        ws = _lib.workspace(x.device, lib.ud_bn_act_workspace_bytes(C), "bn_act")
        _lib.check(lib.ud_bn_act_bwd(_lib.ptr(x), _lib.ptr(y), _lib.ptr(dy), _lib.ptr(scale), _lib.ptr(shift),
                                     _lib.ptr(mean), _lib.ptr(invstd), _lib.ptr(dx), _lib.ptr(dres),
                                     _lib.ptr(dgb[0]), _lib.ptr(dgb[1]), P, C, 1 if relu else 0,
                                     _lib.ptr(ws), ws.numel(), _lib.stream_of(x)), "ud_bn_act_bwd")
        if has_res and ctx.needs_input_grad[3] and dres is None:
            dres = dy
        return dx, dgb[0], dgb[1], dres, None, None, None, None, None, None, None


def bn_act(bn, x, residual=None, relu=True):
    """relu(bn(x) + residual) with nn.BatchNorm2d ``bn``'s parameters, buffers and train/eval mode."""
    tracked = None
    if bn.training:
        nbt = bn.num_batches_tracked
        if nbt.is_cuda and nbt.dtype == torch.int64 and bn.running_mean.dtype == torch.float32:
            tracked = nbt                     # incremented by the statistics kernel (no extra launch)
        else:
            nbt.add_(1)
    return _BnActFn.apply(x, bn.weight, bn.bias, residual, bn.running_mean, bn.running_var, bn.training,
                          bn.momentum, bn.eps, relu, tracked)
