"""3x3 / stride 1 / pad 1 convolution on channels-last bf16 tensors (ud_conv3x3_nhwc_bf16).

Stands in for nn.Conv2d(k=3, s=1, p=1) in the BEV trunk and the detection head of the reference
(unidistill/layers/blocks_2d/det3d/base_bev_backbone.py:30-110, layers/head/det3d/center_head.py:408-420)
in the bf16 mixed-precision mode.  Forward, data gradient and weight gradient run on the hand-written MFMA
kernels.  ``conv1x1`` covers the 1x1 / stride-1 convolutions of the ResNet bottlenecks: y and dx come from
ud_conv1x1_nhwc_bf16 on maps above 1 k pixels and from the library GEMM below, the weight gradient -- a
pixel-reduced GEMM that BLAS libraries run on a handful of CUs -- is ud_conv1x1_wgrad_nhwc_bf16.
"""
import ctypes

import torch

from .. import _lib
from . import bn_act, wgrad_stream


def supported(x, weight, stride=1, padding=1, dilation=1, groups=1):
    def one(v):
        return v[0] if isinstance(v, (tuple, list)) else v
    return (x.is_cuda and x.dtype == torch.bfloat16 and x.dim() == 4 and weight.dim() == 4
            and tuple(weight.shape[2:]) == (3, 3) and one(stride) == 1 and one(padding) == 1
            and one(dilation) == 1 and groups == 1 and weight.shape[1] % 64 == 0
            and weight.shape[0] % 64 == 0)


def _nhwc(t):
    return t if t.is_contiguous(memory_format=torch.channels_last) else \
        t.contiguous(memory_format=torch.channels_last)


FLOP_COUNTER = None      # set to [0] to accumulate the multiply-add count of every kernel launch (bench.py)
SHAPE_LOG = None         # set to [] to record (B, Cin, H, W, Cout, reverse_taps) of every 3x3 launch (bench.py replays them)


def _bn_partial(nbytes, dev):
    """Buffer for the per-tile BatchNorm partials a ``*_bnstats`` convolution writes + the host int it reports."""
    return torch.empty(max(int(nbytes) // 4, 1), dtype=torch.float32, device=dev), ctypes.c_int(0)


def _launch(x, w_tap, cout, bias=None, scale=None, shift=None, residual=None, relu=False, reverse_taps=False,
            bn_stats=False):
    """x: [B, Cin, H, W] bf16 channels-last; w_tap: [Cout, 3, 3, Cin] bf16 contiguous.  bn_stats: also return the
    per-tile (sum, sum of squares) of the output for the BatchNorm that follows: (y, (partial, slices, rows))."""
    B, cin, H, W = x.shape
    if FLOP_COUNTER is not None:
        FLOP_COUNTER[0] += 2 * B * H * W * cout * 9 * cin
    if SHAPE_LOG is not None:
        SHAPE_LOG.append((B, cin, H, W, cout, bool(reverse_taps)))
    y = torch.empty((B, cout, H, W), dtype=torch.bfloat16, device=x.device,
                    memory_format=torch.channels_last)
    if bn_stats:
        lib = _lib.load()
        part, ns = _bn_partial(lib.ud_conv3x3_bnstats_bytes(B, H, W, cout), x.device)
        _lib.check(lib.ud_conv3x3_bnstats_nhwc_bf16(_lib.ptr(x), _lib.ptr(w_tap), _lib.ptr(y), B, H, W, cin, cout,
                                                    _lib.ptr(bias), _lib.ptr(part), part.numel() * 4,
                                                    ctypes.addressof(ns), _lib.stream_of(x)),
                   "ud_conv3x3_bnstats_nhwc_bf16")
        return y, (part, ns.value, B * H * W)
    _lib.check(_lib.load().ud_conv3x3_nhwc_bf16(_lib.ptr(x), _lib.ptr(w_tap), _lib.ptr(y), B, H, W, cin,
                                                cout, _lib.ptr(bias), _lib.ptr(scale), _lib.ptr(shift),
                                                _lib.ptr(residual), (1 if relu else 0) | (2 if reverse_taps else 0),
                                                _lib.stream_of(x)), "ud_conv3x3_nhwc_bf16")
    return y


def _cached(weight, key, make):
    """bf16 re-layouts of a FROZEN weight (requires_grad False: the distillation teacher) are cached on
    the tensor and rebuilt when its version counter moves.  Trainable weights are converted on every
    call: fused optimizers update parameters without touching the version counter."""
    if weight.requires_grad or torch.cuda.is_current_stream_capturing():
        if weight.requires_grad and getattr(weight, key, None) is not None:
            setattr(weight, key, None)      # trained now, possibly frozen again later: never serve the old copy
        return make(weight.detach())
    ver = (weight._version, weight.data_ptr())
    hit = getattr(weight, key, None)
    if hit is None or hit[0] != ver:
        with torch.no_grad():
            hit = (ver, make(weight.detach()))
        try:
            setattr(weight, key, hit)
        except AttributeError:          # non-leaf views etc.: just do not cache
            pass
    return hit[1]


def tap_major(weight):
    """[Cout, Cin, 3, 3] -> [Cout, 3, 3, Cin] bf16 contiguous (the kernel's weight layout)."""
    def make(w):      # permute + cast in ONE copy kernel
        return torch.empty((w.shape[0], 3, 3, w.shape[1]), dtype=torch.bfloat16, device=w.device).copy_(
            w.permute(0, 2, 3, 1))
    return _cached(weight, "_ud_tap", make)


def tap_major_transposed(weight):
    """Weights of the data-gradient convolution, [Cin, 3, 3, Cout] (NOT flipped: the kernel walks the
    taps in reverse, ``reverse_taps``)."""
    def make(w):
        return torch.empty((w.shape[1], 3, 3, w.shape[0]), dtype=torch.bfloat16, device=w.device).copy_(
            w.permute(1, 2, 3, 0))
    return _cached(weight, "_ud_tap_t", make)


def library_layout(weight):
    """bf16 channels-last copy for aten.convolution_backward (weight gradient)."""
    return _cached(weight, "_ud_cl",
                   lambda w: w.to(torch.bfloat16).contiguous(memory_format=torch.channels_last))


USE_HIP_WGRAD = "auto"     # True / False / "auto": hand-written weight-gradient kernel vs aten.convolution_backward
                           # (auto: ours wherever the kernel applies -- 1.2-1.6x the library on every conv shape
                           # of the step, tools/time_conv2d.py -- and deterministic, which the library's is not)


def weight_grad(x, gy, weight):
    """dL/dweight [Cout, Cin, 3, 3] from channels-last bf16 x and gy."""
    cout, cin = weight.shape[0], weight.shape[1]
    if USE_HIP_WGRAD is True or (USE_HIP_WGRAD == "auto" and cin % 64 == 0 and cout % 8 == 0):
        lib = _lib.load()
        B, _, H, W = x.shape
        need = lib.ud_conv3x3_wgrad_workspace_bytes(B, H, W, cin, cout)
        ws = _lib.workspace(x.device, need, "conv_wgrad")
        dw = torch.empty((cout, 3, 3, cin), dtype=torch.float32, device=x.device)
        _lib.check(lib.ud_conv3x3_wgrad_nhwc_bf16(_lib.ptr(x), _lib.ptr(gy), _lib.ptr(dw), B, H, W, cin, cout,
                                                  _lib.ptr(ws), ws.numel(), _lib.stream_of(x)),
                   "ud_conv3x3_wgrad_nhwc_bf16")
        return dw.permute(0, 3, 1, 2).to(weight.dtype)
    if USE_HIP_WGRAD == "auto":
        _lib.library_fallthrough("ops.conv2d.weight_grad (3x3 bf16)", x, gy, weight)
    wb = library_layout(weight)
    return torch.ops.aten.convolution_backward(gy, x, wb, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1,
                                               [False, True, False])[1].to(weight.dtype)


class _Conv3x3Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, holder=None):
        _lib.require_gpu(x, weight)
        x = _nhwc(x)
        y = _launch(x, tap_major(weight), weight.shape[0],
                    None if bias is None else bias.detach().float().contiguous(), bn_stats=holder is not None)
        if holder is not None:
            y, part = y
            holder.append(part)
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        return y

    @staticmethod
    def backward(ctx, gy):
        x, weight = ctx.saved_tensors
        gy = _nhwc(gy.to(torch.bfloat16))
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            gx = _launch(gy, tap_major_transposed(weight), weight.shape[1], reverse_taps=True)
        if ctx.needs_input_grad[1]:
            gw = wgrad_stream.defer(weight, lambda: weight_grad(x, gy, weight), x, gy)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            gb = bn_act.bias_grad(gy)
        return gx, gw, gb, None


def _with_partial(y, holder):
    """Hang the BatchNorm partials of a ``bn_stats`` convolution on its output (ops.bn_act.bn_act picks them up)."""
    out = y[0] if isinstance(y, tuple) else y
    if holder:
        out._ud_bn_partial = holder[0]
    return y


def conv3x3(x, weight, bias=None, bn_stats=False):
    """y = conv2d(x, weight, bias, stride=1, padding=1) for bf16 channels-last x (autograd-aware).  bn_stats: a
    training-mode BatchNorm follows -- its statistics pass is folded into the convolution's epilogue."""
    if not bn_stats:
        return _Conv3x3Fn.apply(x, weight, bias)
    holder = []
    return _with_partial(_Conv3x3Fn.apply(x, weight, bias, holder), holder)


def conv3x3_inference(x, weight, bias=None, scale=None, shift=None, residual=None, relu=False):
    """Inference conv with the fused epilogue: (+bias) (*scale+shift) (+residual) (ReLU)."""
    with torch.no_grad():
        return _launch(_nhwc(x), tap_major(weight), weight.shape[0],
                       None if bias is None else bias.float().contiguous(), scale, shift,
                       None if residual is None else _nhwc(residual), relu)


# ---- 1x1 / stride 1 ------------------------------------------------------------------------------------
GEMM_1X1_MAX_PIXELS = 1024     # per image; above this MIOpen's 1x1 conv beats the GEMM (tools/exp_conv1x1.py)


def supported_1x1(x, weight, strided=False):
    """strided: the 1x1 / stride s path, whose data gradient reduces over Cout in 64- (bf16) / 32-channel (fp32) slices."""
    return (x.is_cuda and x.dim() == 4 and weight.dim() == 4 and tuple(weight.shape[2:]) == (1, 1)
            and weight.shape[1] % 64 == 0 and weight.shape[0] % (64 if strided else 8) == 0)


def weight_grad_1x1(x, gy, weight):
    """dL/dweight [Cout, Cin, 1, 1] (fp32) from channels-last bf16 x [B,Cin,H,W] and gy [B,Cout,H,W]."""
    cout, cin = weight.shape[0], weight.shape[1]
    lib = _lib.load()
    P = x.shape[0] * x.shape[2] * x.shape[3]
    need = lib.ud_conv1x1_wgrad_workspace_bytes(P, cin, cout)
    ws = _lib.workspace(x.device, need, "conv_wgrad")
    dw = torch.empty((cout, cin, 1, 1), dtype=torch.float32, device=x.device)
    _lib.check(lib.ud_conv1x1_wgrad_nhwc_bf16(_lib.ptr(x), _lib.ptr(gy), _lib.ptr(dw), P, cin, cout,
                                              _lib.ptr(ws), ws.numel(), _lib.stream_of(x)),
               "ud_conv1x1_wgrad_nhwc_bf16")
    return dw.to(weight.dtype)


def _w1x1(weight):
    return _cached(weight, "_ud_1x1", lambda w: w.to(torch.bfloat16).contiguous(memory_format=torch.channels_last))


def _w1x1_t(weight):
    """[Cin, Cout] bf16: the weights of the data-gradient 1x1 convolution."""
    return _cached(weight, "_ud_1x1_t",
                   lambda w: torch.empty((w.shape[1], w.shape[0]), dtype=torch.bfloat16, device=w.device).copy_(
                       w.view(w.shape[0], w.shape[1]).t()))


def _w1x1_t_padded(weight):
    """[Cin, Cout rounded up to 64] bf16, zero columns past Cout: the data-gradient weights when Cout is not a multiple of the
    kernel's 64-channel reduction slices (cached for frozen weights like every other re-layout)."""
    def make(w):
        cout, cin = w.shape[0], w.shape[1]
        wt = torch.zeros((cin, cout + (-cout) % 64), dtype=torch.bfloat16, device=w.device)
        wt[:, :cout] = w.view(cout, cin).t()
        return wt
    return _cached(weight, "_ud_1x1_tp", make)


HIP_1X1_MIN_PIXELS = 1         # per image: maps at least this large run the hand-written 1x1 kernel (forward and data
                               # gradient).  With the straight-line slice loop it is within 3 us of the library GEMM on the
                               # 8 x 22 / 16 x 44 maps and ahead everywhere else (tools/time_conv1x1.py), so every 1x1 runs
                               # on it; the GEMM / library branches below only serve channel counts the kernel rejects


def _launch1x1(x, w2d, cout, bias=None, scale=None, shift=None, residual=None, relu=False, bn_stats=False):
    """x: [B, Cin, H, W] bf16 channels-last; w2d: [Cout, Cin] bf16 contiguous."""
    B, cin, H, W = x.shape
    y = torch.empty((B, cout, H, W), dtype=torch.bfloat16, device=x.device, memory_format=torch.channels_last)
    if bn_stats:
        lib = _lib.load()
        part, ns = _bn_partial(lib.ud_conv1x1_bnstats_bytes(B * H * W, cout), x.device)
        _lib.check(lib.ud_conv1x1_bnstats_nhwc_bf16(_lib.ptr(x), _lib.ptr(w2d), _lib.ptr(y), B * H * W, cin, cout,
                                                    _lib.ptr(bias), _lib.ptr(part), part.numel() * 4,
                                                    ctypes.addressof(ns), _lib.stream_of(x)),
                   "ud_conv1x1_bnstats_nhwc_bf16")
        return y, (part, ns.value, B * H * W)
    _lib.check(_lib.load().ud_conv1x1_nhwc_bf16(_lib.ptr(x), _lib.ptr(w2d), _lib.ptr(y), B * H * W, cin, cout,
                                                _lib.ptr(bias), _lib.ptr(scale), _lib.ptr(shift),
                                                _lib.ptr(residual), 1 if relu else 0, _lib.stream_of(x)),
               "ud_conv1x1_nhwc_bf16")
    return y


class _Conv1x1Fn(torch.autograd.Function):
    """with_skip: the function also returns its input (an alias) for the caller's identity branch, so that the
    gradient arriving through that branch is added in the data-gradient kernel's epilogue (or the GEMM's
    beta term) instead of by a separate elementwise add of two full-size tensors (ResNet residual joins)."""

    @staticmethod
    def forward(ctx, x, weight, bias, with_skip=False, holder=None):
        _lib.require_gpu(x, weight)
        x = _nhwc(x)
        wb = _w1x1(weight)
        B, cin, H, W = x.shape
        cout = weight.shape[0]
        ctx.gemm = H * W <= GEMM_1X1_MAX_PIXELS
        ctx.hip = H * W >= HIP_1X1_MIN_PIXELS
        if ctx.hip:
            y = _launch1x1(x, wb.view(cout, cin), cout, None if bias is None else bias.detach().float().contiguous(),
                           bn_stats=holder is not None)
            if holder is not None:
                y, part = y
                holder.append(part)
        elif ctx.gemm:
            _lib.library_fallthrough("ops.conv2d._Conv1x1Fn.forward (GEMM)", x, weight)
            # a 1x1 convolution of a channels-last map IS a plain GEMM [pixels, Cin] x [Cin, Cout]: on the
            # small maps of the deep ResNet stages the library GEMM is 1.2-2.5x faster than the conv solver
            y = x.permute(0, 2, 3, 1).reshape(B * H * W, cin) @ wb.view(cout, cin).t()
            if bias is not None:
                y = y + bias.to(torch.bfloat16)
            y = y.view(B, H, W, cout).permute(0, 3, 1, 2)
        else:
            _lib.library_fallthrough("ops.conv2d._Conv1x1Fn.forward", x, weight)
            y = torch.nn.functional.conv2d(x, wb, None if bias is None else bias.to(torch.bfloat16))
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        return (y, x) if with_skip else y

    @staticmethod
    def backward(ctx, gy, gskip=None):
        x, weight = ctx.saved_tensors
        gy = _nhwc(gy.to(torch.bfloat16))
        B, cin, H, W = x.shape
        cout = weight.shape[0]
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            wb = _w1x1(weight)
            if gskip is not None:
                gskip = _nhwc(gskip.to(torch.bfloat16))
            if ctx.hip and cout % 64 == 0:       # the kernel's reduction dimension comes in 64-channel slices
                gx = _launch1x1(gy, _w1x1_t(weight), cin, residual=gskip)
            elif ctx.hip:
                # Cout not a multiple of 64 (the depth net, 512 -> 368): zero-pad dy and the transposed weights to the next
                # multiple (one copy of dy) instead of leaving the hand-written path for a library GEMM
                pad = (-cout) % 64
                gyp = torch.empty((B, cout + pad, H, W), dtype=torch.bfloat16, device=gy.device,
                                  memory_format=torch.channels_last)
                gyp[:, :cout] = gy
                gyp[:, cout:] = 0
                gx = _launch1x1(gyp, _w1x1_t_padded(weight), cin, residual=gskip)
            elif ctx.gemm:
                _lib.library_fallthrough("ops.conv2d._Conv1x1Fn.backward (GEMM)", gy, weight)
                g2, w2 = gy.permute(0, 2, 3, 1).reshape(B * H * W, cout), wb.view(cout, cin)
                gx = g2 @ w2 if gskip is None else torch.addmm(gskip.permute(0, 2, 3, 1).reshape(B * H * W, cin), g2, w2)
                gx = gx.view(B, H, W, cin).permute(0, 3, 1, 2)
            else:
                _lib.library_fallthrough("ops.conv2d._Conv1x1Fn.backward", gy, weight)
                gx = torch.ops.aten.convolution_backward(gy, x, wb, None, [1, 1], [0, 0], [1, 1], False, [0, 0], 1,
                                                         [True, False, False])[0]
                if gskip is not None:
                    gx = gx + gskip
        if ctx.needs_input_grad[1]:
            gw = wgrad_stream.defer(weight, lambda: weight_grad_1x1(x, gy, weight), x, gy)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            gb = bn_act.bias_grad(gy)
        return gx, gw, gb, None, None


def conv1x1(x, weight, bias=None, bn_stats=False):
    """y = conv2d(x, weight, bias) for a 1x1 / stride-1 convolution of a bf16 channels-last map."""
    if not bn_stats:
        return _Conv1x1Fn.apply(x, weight, bias)
    holder = []
    return _with_partial(_Conv1x1Fn.apply(x, weight, bias, False, holder), holder)


def conv1x1_skip(x, weight, bias=None, bn_stats=False):
    """(conv1x1(x), x): use the second value for the identity branch of a residual block (see _Conv1x1Fn)."""
    if not bn_stats:
        return _Conv1x1Fn.apply(x, weight, bias, True)
    holder = []
    return _with_partial(_Conv1x1Fn.apply(x, weight, bias, True, holder), holder)


# ---- convolutions whose im2col is a permutation: k = s / stride s, transposed k = s / stride s, 1x1 / stride s ----
# (SECONDFPN levels of the image neck, BaseBEVBackbone's up-sampling deblock -- base_bev_backbone.py:67-92 --, the
#  stride-2 shortcut convs of the ResNet stages).  All three passes run on the 1x1 MFMA kernels through a pixel map
#  (ud_conv1x1_mapped_nhwc_bf16 / ud_conv1x1_wgrad_mapped_nhwc_bf16); no im2col buffer, no library call.
import ctypes as _ct


def _pmap(mode, s, Ho, Wo, H, W, C, a=0, b=0):
    return (_ct.c_int * 9)(mode, s, Ho, Wo, H, W, C, a, b)


def _mapped(x, w2d, y, P, K, N, imap, omap):
    """1x1 kernel over pixel maps; bf16 or (the reference's arithmetic) fp32 by the dtype of x."""
    if x.dtype == torch.float32:
        from . import conv2d_f32 as _c32
        if _c32.LOG_1X1 is not None:
            _c32.LOG_1X1.append(("mapped", P, K, N, None if imap is None else tuple(imap), None if omap is None else tuple(omap),
                                 tuple(x.shape), tuple(y.shape)))
        if _c32.persistent_1x1(K, mapped=True, P=P, N=N) and x.numel() < (1 << 30) - (1 << 18) and y.numel() < (1 << 30) - (1 << 18):
            _c32.launch_1x1p(x, w2d, y, P, K, N, imap=imap, omap=omap)
            return
        _lib.check(_lib.load().ud_conv1x1_mapped_nhwc_f32(_lib.ptr(x), _lib.ptr(w2d), _lib.ptr(y), P, K, N, imap, omap,
                                                          _lib.stream_of(x)), "ud_conv1x1_mapped_nhwc_f32")
        return
    _lib.check(_lib.load().ud_conv1x1_mapped_nhwc_bf16(_lib.ptr(x), _lib.ptr(w2d), _lib.ptr(y), P, K, N, imap, omap,
                                                       _lib.stream_of(x)), "ud_conv1x1_mapped_nhwc_bf16")


def _cdt(x):
    """Compute dtype of the mapped convolutions: fp32 tensors outside autocast stay fp32 (fp32 MFMA kernels, all three
    passes), everything else runs in bf16."""
    return torch.float32 if (x.dtype == torch.float32 and not torch.is_autocast_enabled("cuda")) else torch.bfloat16


def _mapped_wgrad(x, gy, P, K, N, xmap, ymap):
    """Weight gradient through pixel maps; bf16 or (the reference's arithmetic) fp32 by the dtype of x."""
    if x.dtype == torch.float32:
        from . import conv2d_f32
        return conv2d_f32.wgrad_mapped(x, gy, P, K, N, xmap, ymap)
    lib = _lib.load()
    ws = _lib.workspace(x.device, lib.ud_conv1x1_wgrad_workspace_bytes(P, K, N), "conv_wgrad")
    dw = torch.empty((N, K), dtype=torch.float32, device=x.device)
    _lib.check(lib.ud_conv1x1_wgrad_mapped_nhwc_bf16(_lib.ptr(x), _lib.ptr(gy), _lib.ptr(dw), P, K, N, xmap, ymap,
                                                     _lib.ptr(ws), ws.numel(), _lib.stream_of(x)),
               "ud_conv1x1_wgrad_mapped_nhwc_bf16")
    return dw


def _bf16_cl_empty(shape, dev, zero=False, dtype=torch.bfloat16):
    t = torch.empty(shape, dtype=dtype, device=dev, memory_format=torch.channels_last)
    return t.zero_() if zero else t


def supported_patch(x, weight, s, transposed=False):
    """conv k = s / stride s (weight [Cout, Cin, s, s]) or its transpose (weight [Cin, Cout, s, s])."""
    if not (x.is_cuda and x.dim() == 4 and weight.dim() == 4 and tuple(weight.shape[2:]) == (s, s) and s >= 2):
        return False
    cin, cout = (weight.shape[0], weight.shape[1]) if transposed else (weight.shape[1], weight.shape[0])
    # the transposed form's data gradient gathers through the mode-1 map: a 64-channel slice must stay inside one block row
    return cin % 64 == 0 and cout % 8 == 0 and x.shape[1] == cin and (transposed or (s * cin) % 64 == 0) \
        and (not transposed or ((s * s * cout) % 64 == 0 and (s * cout) % 64 == 0))


class _ConvPatchFn(torch.autograd.Function):
    """nn.Conv2d(kernel_size=s, stride=s, padding=0, bias=None)."""

    @staticmethod
    def forward(ctx, x, weight, s):
        _lib.require_gpu(x, weight)
        dt = _cdt(x)
        x = _nhwc(x.to(dt))
        B, C, H, W = x.shape
        cout = weight.shape[0]
        Ho, Wo = H // s, W // s
        K, P = s * s * C, B * Ho * Wo
        w2 = _cached(weight, "_ud_patch" + str(dt), lambda w: torch.empty((w.shape[0], s, s, w.shape[1]), dtype=dt,
                                                                            device=w.device).copy_(w.permute(0, 2, 3, 1)))
        y = _bf16_cl_empty((B, cout, Ho, Wo), x.device, dtype=dt)
        _mapped(x, w2, y, P, K, cout, _pmap(1, s, Ho, Wo, H, W, C), None)
        ctx.save_for_backward(x, weight)
        ctx.s = s
        return y

    @staticmethod
    def backward(ctx, gy):
        x, weight = ctx.saved_tensors
        s = ctx.s
        dt = x.dtype
        gy = _nhwc(gy.to(dt))
        B, C, H, W = x.shape
        cout = weight.shape[0]
        Ho, Wo = H // s, W // s
        K, P = s * s * C, B * Ho * Wo
        pm = _pmap(1, s, Ho, Wo, H, W, C)
        gx = gw = None
        if ctx.needs_input_grad[0] and cout % 64 == 0:
            wt = _cached(weight, "_ud_patch_t" + str(dt), lambda w: torch.empty((s, s, w.shape[1], w.shape[0]), dtype=dt,
                                                                                  device=w.device).copy_(w.permute(2, 3, 1, 0)))
            gx = _bf16_cl_empty((B, C, H, W), x.device, zero=(H % s != 0 or W % s != 0), dtype=dt)
            _mapped(gy, wt, gx, P, cout, K, None, pm)          # [P][Cout] x [K][Cout]^T -> rows scattered by the map
        elif ctx.needs_input_grad[0]:
            _lib.library_fallthrough("ops.conv2d._ConvPatchFn.backward", gy, weight, stride=s)
            gx = torch.ops.aten.convolution_backward(gy, x, weight.detach().to(dt), None, [s, s], [0, 0], [1, 1],
                                                     False, [0, 0], 1, [True, False, False])[0]
        if ctx.needs_input_grad[1]:
            gw = wgrad_stream.defer(weight, lambda: _mapped_wgrad(x, gy, P, K, cout, pm, None)    # [Cout][s][s][C]
                                    .view(cout, s, s, C).permute(0, 3, 1, 2).to(weight.dtype), x, gy)
        return gx, gw, None


class _ConvTPatchFn(torch.autograd.Function):
    """nn.ConvTranspose2d(kernel_size=s, stride=s, padding=0, bias=None): every input pixel writes an s x s block."""

    @staticmethod
    def forward(ctx, x, weight, s):
        _lib.require_gpu(x, weight)
        dt = _cdt(x)
        x = _nhwc(x.to(dt))
        B, cin, H, W = x.shape
        cout = weight.shape[1]
        N, P = s * s * cout, B * H * W
        wt = _cached(weight, "_ud_tpatch" + str(dt), lambda w: torch.empty((s, s, w.shape[1], w.shape[0]), dtype=dt,
                                                                             device=w.device).copy_(w.permute(2, 3, 1, 0)))
        y = _bf16_cl_empty((B, cout, H * s, W * s), x.device, dtype=dt)
        _mapped(x, wt, y, P, cin, N, None, _pmap(1, s, H, W, H * s, W * s, cout))
        ctx.save_for_backward(x, weight)
        ctx.s = s
        return y

    @staticmethod
    def backward(ctx, gy):
        x, weight = ctx.saved_tensors
        s = ctx.s
        dt = x.dtype
        gy = _nhwc(gy.to(dt))
        B, cin, H, W = x.shape
        cout = weight.shape[1]
        N, P = s * s * cout, B * H * W
        pm = _pmap(1, s, H, W, H * s, W * s, cout)
        gx = gw = None
        if ctx.needs_input_grad[0]:
            w2 = _cached(weight, "_ud_tpatch_t" + str(dt), lambda w: torch.empty((w.shape[0], s, s, w.shape[1]), dtype=dt,
                                                                                   device=w.device).copy_(w.permute(0, 2, 3, 1)))
            gx = _bf16_cl_empty((B, cin, H, W), x.device, dtype=dt)
            _mapped(gy, w2, gx, P, N, cin, pm, None)            # rows of dy gathered by the map: [P][N] x [Cin][N]^T
        if ctx.needs_input_grad[1]:
            gw = wgrad_stream.defer(weight, lambda: _mapped_wgrad(x, gy, P, cin, N, None, pm)      # [s][s][Cout][Cin]
                                    .view(s, s, cout, cin).permute(3, 2, 0, 1).to(weight.dtype), x, gy)
        return gx, gw, None


class _Conv1x1StrideFn(torch.autograd.Function):
    """nn.Conv2d(kernel_size=1, stride=s, bias=None): the ResNet stage shortcuts."""

    @staticmethod
    def forward(ctx, x, weight, s):
        _lib.require_gpu(x, weight)
        dt = _cdt(x)
        x = _nhwc(x.to(dt))
        B, cin, H, W = x.shape
        cout = weight.shape[0]
        Ho, Wo = (H - 1) // s + 1, (W - 1) // s + 1
        y = _bf16_cl_empty((B, cout, Ho, Wo), x.device, dtype=dt)
        w2 = weight.detach().reshape(cout, cin).contiguous() if dt == torch.float32 else _w1x1(weight).view(cout, cin)
        _mapped(x, w2, y, B * Ho * Wo, cin, cout, _pmap(2, s, Ho, Wo, H, W, cin), None)
        ctx.save_for_backward(x, weight)
        ctx.s = s
        return y

    @staticmethod
    def backward(ctx, gy):
        x, weight = ctx.saved_tensors
        s = ctx.s
        dt = x.dtype
        gy = _nhwc(gy.to(dt))
        B, cin, H, W = x.shape
        cout = weight.shape[0]
        Ho, Wo = (H - 1) // s + 1, (W - 1) // s + 1
        P = B * Ho * Wo
        pm = _pmap(2, s, Ho, Wo, H, W, cin)
        gx = gw = None
        if ctx.needs_input_grad[0]:
            gx = _bf16_cl_empty((B, cin, H, W), x.device, zero=True, dtype=dt)   # only the sampled pixels receive gradient
            wt = weight.detach().reshape(cout, cin).t().contiguous() if dt == torch.float32 else _w1x1_t(weight)
            _mapped(gy, wt, gx, P, cout, cin, None, pm)
        if ctx.needs_input_grad[1]:
            gw = wgrad_stream.defer(weight, lambda: _mapped_wgrad(x, gy, P, cin, cout, pm, None).view(cout, cin, 1, 1).to(weight.dtype),
                                    x, gy)
        return gx, gw, None


def conv_patch(x, weight, s):
    return _ConvPatchFn.apply(x, weight, s)


def conv_transpose_patch(x, weight, s):
    return _ConvTPatchFn.apply(x, weight, s)


def conv1x1_strided(x, weight, s):
    return _Conv1x1StrideFn.apply(x, weight, s)


class _Conv3x3S2Fn(torch.autograd.Function):
    """nn.Conv2d(kernel_size=3, stride=2, padding=1, bias=None): the first conv of ResNet stages 2-4 and of
    BaseBEVBackbone's second level (ZeroPad2d(1) + unpadded conv there).  Forward and weight gradient read x through
    the im2col map (mode 3); the data gradient runs once per input-pixel parity class (mode 4 gather from dy, mode 2
    scatter into dx): exactly the convolution's multiply-adds in every pass, no zero-stuffed taps, no library call."""

    @staticmethod
    def forward(ctx, x, weight):
        _lib.require_gpu(x, weight)
        dt = _cdt(x)
        x = _nhwc(x.to(dt))
        B, C, H, W = x.shape
        cout = weight.shape[0]
        Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
        y = _bf16_cl_empty((B, cout, Ho, Wo), x.device, dtype=dt)
        wt = weight.detach().permute(0, 2, 3, 1).contiguous() if dt == torch.float32 else tap_major(weight)
        _mapped(x, wt, y, B * Ho * Wo, 9 * C, cout, _pmap(3, 2, Ho, Wo, H, W, C), None)
        ctx.save_for_backward(x, weight)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, weight = ctx.saved_tensors
        dt = x.dtype
        gy = _nhwc(gy.to(dt))
        B, C, H, W = x.shape
        cout = weight.shape[0]
        Ho, Wo = gy.shape[2], gy.shape[3]
        gx = gw = None
        if ctx.needs_input_grad[0]:
            # [C][3][3][Cout]: the kernel picks each class's taps from it
            wtt = weight.detach().permute(1, 2, 3, 0).contiguous() if dt == torch.float32 else tap_major_transposed(weight)
            gx = _bf16_cl_empty((B, C, H, W), x.device, dtype=dt)
            for a in (0, 1):
                for b in (0, 1):
                    Hc, Wc = (H - a + 1) // 2, (W - b + 1) // 2
                    if Hc <= 0 or Wc <= 0:
                        continue
                    K = (1 + a) * (1 + b) * cout
                    _mapped(gy, wtt, gx, B * Hc * Wc, K, C, _pmap(4, 2, Hc, Wc, Ho, Wo, cout, a, b),
                            _pmap(2, 2, Hc, Wc, H, W, C, a, b))
        if ctx.needs_input_grad[1]:
            gw = wgrad_stream.defer(weight, lambda: _mapped_wgrad(x, gy, B * Ho * Wo, 9 * C, cout, _pmap(3, 2, Ho, Wo, H, W, C), None)   # [Cout][9][C]
                                    .view(cout, 3, 3, C).permute(0, 3, 1, 2).to(weight.dtype), x, gy)
        return gx, gw


def supported_3x3_s2(x, weight):
    return (x.is_cuda and x.dim() == 4 and weight.dim() == 4 and tuple(weight.shape[2:]) == (3, 3)
            and weight.shape[1] % 64 == 0 and weight.shape[0] % 64 == 0 and x.shape[1] == weight.shape[1])


def conv3x3_stride2(x, weight):
    return _Conv3x3S2Fn.apply(x, weight)
