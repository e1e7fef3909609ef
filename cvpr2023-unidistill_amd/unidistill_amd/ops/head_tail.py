"""BatchNorm -> ReLU -> per-head 3x3 conv for the packed detection heads, on libunidistill_hip.

Replaces modules 1..3 of every SepHead stack of the reference
(unidistill/layers/head/det3d/center_head.py:311-362) once the first convs are packed into one
(layers/center_head.py: PackedSepHeads).  Entry points: ud_head_tail_{stats,fwd,bwd}.
"""
import torch

from .. import _lib

HIDDEN = 64      # SepHead head_conv the kernels are built for
MAX_OUT = 3      # widest head the MFMA packing holds (3 x 9 taps <= 32)


def supported(y, head_conv, kmax, kernel):
    return (y.is_cuda and y.dtype == torch.bfloat16 and head_conv == HIDDEN and kmax <= MAX_OUT
            and kernel == 3 and y.dim() == 4)


def _nhwc(y):
    """[B, C, H, W] tensor -> the same tensor with channels-last storage (no copy when it already is)."""
    return y if y.is_contiguous(memory_format=torch.channels_last) else \
        y.contiguous(memory_format=torch.channels_last)


def _weights_tap_major(w2, G, kmax):
    """packed [G*kmax, 64, 3, 3] -> [G, kmax, 9, 64] fp32 (the kernels' layout)."""
    return w2.detach().float().view(G, kmax, HIDDEN, 9).permute(0, 1, 3, 2).contiguous()


class _HeadTailFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, y, gamma, beta, w2, b2, running_mean, running_var, training, momentum, eps,
                G, kmax):
        lib = _lib.load()
        _lib.require_gpu(y, gamma, beta, w2, b2)
        y = _nhwc(y)
        B, C, H, W = y.shape
        assert C == G * HIDDEN
        dev = y.device
        wk = _weights_tap_major(w2, G, kmax)
        g32, b32 = gamma.detach().float().contiguous(), beta.detach().float().contiguous()
        ws = _lib.workspace(dev, lib.ud_head_tail_workspace_bytes(G), "head_tail")
        stream = _lib.stream_of(y)
        if training:
            vec = torch.empty((5, C), dtype=torch.float32, device=dev)
            mean, var, invstd, scale, shift = vec[0], vec[1], vec[2], vec[3], vec[4]
            fp32_buffers = running_mean is not None and running_mean.dtype == torch.float32
            rm, rv = (running_mean, running_var) if fp32_buffers else (None, None)   # updated in-kernel
            _lib.check(lib.ud_head_tail_stats(_lib.ptr(y), B, H, W, G, _lib.ptr(g32), _lib.ptr(b32),
                                              float(eps), _lib.ptr(mean), _lib.ptr(var),
                                              _lib.ptr(invstd), _lib.ptr(scale), _lib.ptr(shift), _lib.ptr(rm), _lib.ptr(rv),
                                              float(momentum or 0.0), None, _lib.ptr(ws), ws.numel(), stream), "ud_head_tail_stats")
            if rm is not None:            # written through raw pointers by the kernel: move the version counters
                torch.autograd.graph.increment_version(rm)
                torch.autograd.graph.increment_version(rv)
            if running_mean is not None and not fp32_buffers:
                n = B * H * W
                with torch.no_grad():     # nn.BatchNorm2d bookkeeping: unbiased variance in the buffers
                    running_mean.mul_(1 - momentum).add_(mean, alpha=momentum)
                    running_var.mul_(1 - momentum).add_(var, alpha=momentum * n / max(n - 1, 1))
        else:
            invstd = torch.rsqrt(running_var.float() + eps)
            mean = running_mean.float()
            scale = (g32 * invstd).contiguous()
            shift = (b32 - mean * scale).contiguous()
        z = torch.empty((B, G * kmax, H, W), dtype=torch.float32, device=dev)
        _lib.check(lib.ud_head_tail_fwd(_lib.ptr(y), _lib.ptr(scale), _lib.ptr(shift), _lib.ptr(wk),
                                        _lib.ptr(b2.detach().float().contiguous()), _lib.ptr(z),
                                        B, H, W, G, kmax, stream), "ud_head_tail_fwd")
        ctx.save_for_backward(y, wk, scale, shift, mean, invstd)
        ctx.cfg = (bool(training), G, kmax, w2.shape)
        return z

    @staticmethod
    def backward(ctx, dz):
        y, wk, scale, shift, mean, invstd = ctx.saved_tensors
        training, G, kmax, wshape = ctx.cfg
        if not training:
            raise NotImplementedError("head tail backward is implemented for training-mode BatchNorm "
                                      "(the reference never back-propagates through an eval-mode head)")
        lib = _lib.load()
        B, C, H, W = y.shape
        dz = dz.contiguous().float()
        dy = torch.empty_like(y)                      # channels-last bf16 like y
        dwk = torch.empty_like(wk)
        dgb = torch.empty((2, C), dtype=torch.float32, device=y.device)
        ws = _lib.workspace(y.device, lib.ud_head_tail_workspace_bytes(G), "head_tail")
        _lib.check(lib.ud_head_tail_bwd(_lib.ptr(y), _lib.ptr(dz), _lib.ptr(wk), _lib.ptr(scale),
                                        _lib.ptr(shift), _lib.ptr(mean), _lib.ptr(invstd), _lib.ptr(dy),
                                        _lib.ptr(dwk), _lib.ptr(dgb[0]), _lib.ptr(dgb[1]), B, H, W, G,
                                        kmax, _lib.ptr(ws), ws.numel(), _lib.stream_of(y)),
                   "ud_head_tail_bwd")
        dw2 = dwk.permute(0, 1, 3, 2).reshape(wshape)
        db2 = dz.sum((0, 2, 3))
        return dy, dgb[0], dgb[1], dw2, db2, None, None, None, None, None, None, None


def head_tail(y, gamma, beta, w2, b2, running_mean, running_var, training, momentum, eps, G, kmax):
    """z[B, G*kmax, H, W] (fp32) = conv3x3_per_head(relu(batch_norm(y)), w2) + b2.

    y: [B, G*64, H, W] bf16 (channels-last storage preferred); w2: [G*kmax, 64, 3, 3]; running
    statistics are updated in place when ``training`` (momentum as in nn.BatchNorm2d)."""
    return _HeadTailFn.apply(y, gamma, beta, w2, b2, running_mean, running_var, training, momentum,
                             eps, G, kmax)
