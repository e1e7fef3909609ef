"""BatchNorm2d (+ residual) (+ ReLU) on channels-last bf16 activations (ud_bn_act_*, ud_head_tail_stats).

The BatchNorm -> ReLU links of the reference's dense layers (base_bev_backbone.py:48-66,
center_head.py:408-420, mmdet ResNet bottlenecks) as one statistics pass + one streaming pass; the
backward recomputes the ReLU mask from x instead of storing it.
"""
import torch

from .. import _lib


def supported(x, bn):
    """4-D channels-last maps [B, C, H, W] or 2-D row tensors [M, C] (sparse voxel features); bf16 (mixed
    precision mode) or fp32 (the reference's arithmetic)."""
    if not (x.is_cuda and x.dtype in (torch.bfloat16, torch.float32) and x.dim() in (2, 4) and x.shape[1] % 16 == 0
            and x.shape[0] > 0 and bn.affine and bn.track_running_stats and bn.momentum is not None):
        return False
    return x.is_contiguous() if x.dim() == 2 else x.is_contiguous(memory_format=torch.channels_last)


def _like(t, ref):
    """t in ref's dtype/layout (contiguous rows or channels-last map)."""
    t = t.to(ref.dtype)
    return t.contiguous() if ref.dim() == 2 else t.contiguous(memory_format=torch.channels_last)


def _row_stride(t, ref):
    """t [B, C, H, W] addressed as rows of C channels every `ld` elements -- a channel slice of a (wider) channels-last map, e.g.
    what torch.cat's backward hands to each input: -> ld (== C for a plain channels-last tensor), or None."""
    if t.dim() != 4 or ref.dim() != 4 or t.dtype != ref.dtype or t.shape != ref.shape or not t.is_cuda:
        return None
    B, C, H, W = t.shape
    s = t.stride()
    ld = s[3] if W > 1 else (s[2] if H > 1 else C)
    if ld < C or ld % 8 or t.data_ptr() % 16 or (C > 1 and s[1] != 1) or (W > 1 and s[3] != ld) or (H > 1 and s[2] != W * ld) \
            or (B > 1 and s[0] != H * W * ld):
        return None
    return ld


_ws_bytes = {}


def _workspace(dev, C):
    n = _ws_bytes.get(C)
    if n is None:
        n = _ws_bytes[C] = _lib.load().ud_bn_act_workspace_bytes(C)
    return _lib.workspace(dev, n, "bn_act")


def batch_stats(x, gamma, beta, running_mean, running_var, training, momentum, eps, tracked=None, partial=None):
    """-> f32 [5, C]: mean, var, invstd, scale, shift of BatchNorm over x ([B, C, H, W] channels-last or [M, C]); training: batch
    statistics (from the producing convolution's per-tile partials when given) and the running buffers updated; eval: running."""
    lib = _lib.load()
    _lib.require_gpu(x, gamma, beta)
    f32 = x.dtype == torch.float32
    k_stats = lib.ud_bn_stats_f32 if f32 else lib.ud_bn_stats
    C = x.shape[1]
    P = x.numel() // C
    dev = x.device
    g32 = gamma.detach() if gamma.dtype == torch.float32 else gamma.detach().float()
    b32 = beta.detach() if beta.dtype == torch.float32 else beta.detach().float()
    stream = _lib.stream_of(x)
    if training:
        # one [5, C] block: mean, var, invstd, scale, shift (addressed by offset: no per-row views)
        vec = torch.empty((5, C), dtype=torch.float32, device=dev)
        v0, row = vec.data_ptr(), 4 * C
        ws = _workspace(dev, C)
        fp32_buffers = running_mean is not None and running_mean.dtype == torch.float32
        rm, rv = (running_mean, running_var) if fp32_buffers else (None, None)   # updated in-kernel
        if partial is not None and partial[2] == P:
            # the producing convolution already reduced every tile (ud_conv*_bnstats_nhwc_*): second pass only
            _lib.check(lib.ud_bn_stats_from_partials(partial[0].data_ptr(), int(partial[1]), P, C, g32.data_ptr(),
                                                     b32.data_ptr(), float(eps), v0, v0 + row, v0 + 2 * row,
                                                     v0 + 3 * row, v0 + 4 * row, _lib.ptr(rm), _lib.ptr(rv),
                                                     float(momentum or 0.0), _lib.ptr(tracked), stream),
                       "ud_bn_stats_from_partials")
        else:
            _lib.check(k_stats(x.data_ptr(), P, C, g32.data_ptr(), b32.data_ptr(), float(eps),
                               v0, v0 + row, v0 + 2 * row, v0 + 3 * row, v0 + 4 * row,
                               _lib.ptr(rm), _lib.ptr(rv), float(momentum or 0.0),
                               _lib.ptr(tracked), ws.data_ptr(), ws.numel(), stream), "ud_bn_stats")
        if rm is not None:       # written through raw pointers by the statistics kernel: tell autograd's version counters
            torch.autograd.graph.increment_version(rm)
            torch.autograd.graph.increment_version(rv)
        if running_mean is not None and not fp32_buffers:
            with torch.no_grad():
                running_mean.mul_(1 - momentum).add_(vec[0], alpha=momentum)
                running_var.mul_(1 - momentum).add_(vec[1], alpha=momentum * P / max(P - 1, 1))
        return vec
    # eval mode: the folded vectors depend on parameters / buffers only -- for FROZEN affine parameters (the distillation teacher:
    # ~70 eval-mode BatchNorms per step, six tiny launches each) they are computed once per version of the four tensors.  Trainable
    # gamma / beta are folded on every call (fused optimizers do not move the version counter); the kernels that update the running
    # buffers in place bump their versions (see above).
    def fold():
        invstd = torch.rsqrt(running_var.float() + eps)
        mean = running_mean.float()
        scale = g32 * invstd
        return torch.stack((mean, invstd * invstd, invstd, scale, b32 - mean * scale))

    if gamma.requires_grad or beta.requires_grad or torch.cuda.is_current_stream_capturing():
        if getattr(running_var, "_ud_bn_eval", None) is not None and not torch.cuda.is_current_stream_capturing():
            running_var._ud_bn_eval = None   # trained now, possibly frozen again later: never serve the old fold
        return fold()
    key = (running_mean._version, running_var._version, gamma._version, beta._version, float(eps),
           running_mean.data_ptr(), running_var.data_ptr(), gamma.data_ptr(), beta.data_ptr())
    hit = getattr(running_var, "_ud_bn_eval", None)
    if hit is not None and hit[0] == key:
        return hit[1]
    vec = fold()
    try:
        running_var._ud_bn_eval = (key, vec)
    except (AttributeError, RuntimeError):
        pass
    return vec


def attach_colsum(g, sums):
    """The producer of gradient tensor ``g`` already holds its per-channel sums (emitted by the pass that wrote g): leave them
    on the tensor object for the bias gradient of the convolution that receives g (bias_grad below)."""
    try:
        g._ud_colsum = (g._version, g.data_ptr(), sums)
    except (AttributeError, RuntimeError):
        pass
    return g


def drop_colsum(g):
    """g is about to be modified in place through a raw pointer (no version bump): its recorded sums no longer hold."""
    if getattr(g, "_ud_colsum", None) is not None:
        g._ud_colsum = None


def bias_grad(gy):
    """f32 [C] = gy.sum over everything but the channel axis -- the bias gradient of a convolution (nn.Conv2d(bias=True):
    center_head.py:64,339,353) -- for a channels-last [B, C, H, W] map or an [M, C] row tensor, fp32 or bf16: ud_colsum_* (two
    HBM-rate passes in a fixed order), or the sums its producer left on the tensor (attach_colsum)."""
    hit = getattr(gy, "_ud_colsum", None)
    if hit is not None and hit[0] == gy._version and hit[1] == gy.data_ptr():
        return hit[2]
    C = gy.shape[1]
    rows = gy.dim() == 2 and gy.is_contiguous()
    maps = gy.dim() == 4 and gy.is_contiguous(memory_format=torch.channels_last)
    if not (gy.is_cuda and gy.dtype in (torch.float32, torch.bfloat16) and (rows or maps) and gy.numel() > 0):
        dims = (0,) if gy.dim() == 2 else (0, 2, 3)
        return gy.sum(dims, dtype=torch.float32)
    lib = _lib.load()
    out = torch.empty(C, dtype=torch.float32, device=gy.device)
    ws = _lib.workspace(gy.device, lib.ud_colsum_workspace_bytes(C), "colsum")
    fn = lib.ud_colsum_f32 if gy.dtype == torch.float32 else lib.ud_colsum_bf16
    _lib.check(fn(gy.data_ptr(), gy.numel() // C, C, C, out.data_ptr(), ws.data_ptr(), ws.numel(), _lib.stream_of(gy)),
               "ud_colsum")
    return out


class _BnActFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, residual, running_mean, running_var, training, momentum, eps, relu,
                tracked=None, partial=None, out=None):
        lib = _lib.load()
        k_fwd = lib.ud_bn_act_fwd_ld_f32 if x.dtype == torch.float32 else lib.ud_bn_act_fwd_ld
        C = x.shape[1]
        P = x.numel() // C
        stream = _lib.stream_of(x)
        vec = batch_stats(x, gamma, beta, running_mean, running_var, training, momentum, eps, tracked, partial)
        v0, row = vec.data_ptr(), 4 * C
        if residual is not None:
            residual = _like(residual, x)
        # out: a channel slice of a wider channels-last map (a fused concatenation, see cat_slices) written in place
        y = torch.empty_like(x) if out is None else out[0]       # (wrapped in a tuple: not an autograd input of this Function)
        ld = C if out is None else _row_stride(y, x)
        if ld is None:
            raise ValueError(f"bn_act: out {tuple(y.shape)} strides {tuple(y.stride())} is not a channel slice of a channels-last map")
        _lib.check(k_fwd(x.data_ptr(), _lib.ptr(residual), v0 + 3 * row, v0 + 4 * row,
                                     y.data_ptr(), P, C, ld, 1 if relu else 0, stream), "ud_bn_act_fwd_ld")
        ctx.cfg = (bool(training), bool(relu), residual is not None)
        ctx.save_for_backward(x, y if (residual is not None and relu) else None, vec)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y, vec = ctx.saved_tensors
        training, relu, has_res = ctx.cfg
        if not training:
            raise NotImplementedError("fused BatchNorm backward covers training-mode statistics only")
        lib = _lib.load()
        C = x.shape[1]
        P = x.numel() // C
        ld = _row_stride(dy, x)           # a slice of a concatenation's gradient is read in place
        if ld is None:
            dy, ld = _like(dy, x), C
        dx = torch.empty_like(x)
        dres = torch.empty_like(x) if (has_res and ctx.needs_input_grad[3]) else None
        if dres is not None and not relu:
            dres = None                                   # no mask: the residual gradient is dy itself
        dgb = torch.empty((2, C), dtype=torch.float32, device=x.device)
        ws = _workspace(x.device, C)
        v0, row, g0 = vec.data_ptr(), 4 * C, dgb.data_ptr()
        k_bwd = lib.ud_bn_act_bwd_ld_f32 if x.dtype == torch.float32 else lib.ud_bn_act_bwd_ld
        _lib.check(k_bwd(x.data_ptr(), _lib.ptr(y), dy.data_ptr(), ld, v0 + 3 * row, v0 + 4 * row,
                                     v0, v0 + 2 * row, dx.data_ptr(), _lib.ptr(dres),
                                     g0, g0 + row, P, C, 1 if relu else 0,
                                     ws.data_ptr(), ws.numel(), _lib.stream_of(x)), "ud_bn_act_bwd_ld")
        if has_res and ctx.needs_input_grad[3] and dres is None:
            dres = dy
        return dx, dgb[0], dgb[1], dres, None, None, None, None, None, None, None, None, None


class _CatSlices(torch.autograd.Function):
    """The concatenated map whose channel slices the producers have already written (bn_act(..., out=slice)): forward hands the
    buffer on, backward hands each producer ITS slice of the gradient as a strided view (read in place by ud_bn_act_bwd_ld)."""

    @staticmethod
    def forward(ctx, holder, *parts):
        ctx.widths = [p.shape[1] for p in parts]
        return _alias(holder[0], 0, holder[0].size(), holder[0].stride())

    @staticmethod
    def backward(ctx, g):
        if not g.is_contiguous(memory_format=torch.channels_last):
            g = g.contiguous(memory_format=torch.channels_last)
        outs, c0 = [], 0
        for w in ctx.widths:
            outs.append(g[:, c0:c0 + w])
            c0 += w
        return (None, *outs)


def _alias(buf, offset, size, stride):
    """A tensor on buf's storage that autograd does not know as a view of buf (the producers' outputs and the concatenated map
    are separate autograd tensors over one allocation; nothing writes it after the producers)."""
    return torch.empty(0, dtype=buf.dtype, device=buf.device).set_(buf.untyped_storage(), buf.storage_offset() + offset, size, stride)


def cat_buffer(ref, widths):
    """Channels-last [B, sum(widths), H, W] buffer + its channel slices, for bn_act(..., out=slice) and cat_slices."""
    B, _, H, W = ref.shape
    Ct = sum(widths)
    buf = torch.empty((B, H, W, Ct), dtype=ref.dtype, device=ref.device).permute(0, 3, 1, 2)
    slots, c0 = [], 0
    for w in widths:
        slots.append(_alias(buf, c0, (B, w, H, W), buf.stride()))
        c0 += w
    return buf, slots


def cat_slices(buf, parts):
    """== torch.cat(parts, 1) when parts[i] IS the i-th channel slice of buf (written in place by its producer)."""
    return _CatSlices.apply((buf,), *parts)


def bn_act(bn, x, residual=None, relu=True, out=None):
    """relu(bn(x) + residual) with nn.BatchNorm2d ``bn``'s parameters, buffers and train/eval mode; out: see cat_buffer."""
    tracked = None
    if bn.training:
        nbt = bn.num_batches_tracked
        if nbt.is_cuda and nbt.dtype == torch.int64 and bn.running_mean.dtype == torch.float32:
            tracked = nbt                     # incremented by the statistics kernel (no extra launch)
        else:
            nbt.add_(1)
    partial = getattr(x, "_ud_bn_partial", None) if bn.training else None    # left by a bn_stats convolution
    return _BnActFn.apply(x, bn.weight, bn.bias, residual, bn.running_mean, bn.running_var, bn.training,
                          bn.momentum, bn.eps, relu, tracked, partial, None if out is None else (out,))
