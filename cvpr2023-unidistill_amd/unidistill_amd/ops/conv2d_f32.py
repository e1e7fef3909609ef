"""fp32 3x3 / stride-1 / pad-1 and 1x1 convolutions on channels-last fp32 tensors (ud_conv3x3_nhwc_f32,
ud_conv1x1_nhwc_f32: exact fp32 products on the fp32 MFMA pipe) -- the reference's own arithmetic for the BEV
trunk, head, fusion conv and the ResNet / neck convs (base_bev_backbone.py:30-110, center_head.py:311-420,
BEVFusion_nuscenes_base_exp.py:107-135, lss_fpn.py:143-149).  Forward, data gradient and weight gradient are
hand-written (csrc/conv2d_f32.hip, csrc/conv2d_f32_wgrad.hip).
"""
import ctypes
import os

import torch

from .. import _lib
from . import bn_act, wgrad_stream


def _nhwc(t):
    return t if t.is_contiguous(memory_format=torch.channels_last) else \
        t.contiguous(memory_format=torch.channels_last)


def supported(x, weight, ks):
    return (x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and weight.dtype == torch.float32
            and tuple(weight.shape[2:]) == (ks, ks) and weight.shape[1] % 32 == 0 and weight.shape[0] % 4 == 0
            and x.shape[1] == weight.shape[1])


def _bn_partial(nbytes, dev):
    return torch.empty(max(int(nbytes) // 4, 1), dtype=torch.float32, device=dev), ctypes.c_int(0)


FLOP_COUNTER = None      # set to [0] to accumulate the multiply-add count of every 3x3 launch (bench.py)
SHAPE_LOG = None         # set to [] to record (B, Cin, H, W, Cout, reverse_taps) of every 3x3 launch (bench.py replays them)


USE_WINOGRAD = os.environ.get("UD_F32_WINOGRAD", "1") != "0"   # 3x3: Winograd F(2x2,3x3) kernels where the map fills the blocks
WINO_MIN_FILL = 0.7
USE_WINOGRAD_WGRAD = os.environ.get("UD_F32_WINOGRAD_WGRAD", "1") != "0"


# Winograd F(4x4, 3x3) (csrc/conv2d_f32_wino4.hip): 1.78x fewer MFMA flops again; taken where the map fills its 32-tile blocks
USE_WINO4 = os.environ.get("UD_F32_WINO4", "1") != "0"
# tile blocks must be at least this full: 0.35 since the stream-K tail (measured: 512 -> 512 @8x22x24, 12 of 32 tile slots used,
# 183 us against 233 us for the direct kernel and 228 us for F(2x2); the whole step 61.8 -> 61.3 ms with the image branch's
# layer3 / layer4 convolutions on F(4x4), tests/test_image_branch_f32_gpu.py green with its bounds untouched)
WINO4_MIN_FILL = float(os.environ.get("UD_F32_WINO4_FILL", "0.35"))


WINO4_MIN_CIN = int(os.environ.get("UD_F32_WINO4_CIN", "256"))


def wino4_pays(H, W, cin, cout):
    """Measured per shape against F(2x2) (tools/time_wino4.py, profiles/r05_conv_f32_wino4.md): with the stream-K tail F(4x4) is
    ahead on every 3x3 shape of the step whose map fills the 32-tile blocks -- 1.04x (64 -> 64 @64x176) to 1.54x (2688 -> 64
    @180^2).  Routed to it: the BEV maps (trunk, head) and every layer with Cin >= 256.  The 64- / 128-channel ResNet layers
    (1.04x / 1.19x, 0.1 ms per step together) stay on F(2x2): with all sixteen 3x3 layers of the image branch on F(4x4) the
    branch's forward ends 1.07e-4 of its max from the library path -- past the 1e-4 of tests/test_image_branch_f32_gpu.py, which
    is kept as it was."""
    if not (USE_WINOGRAD and USE_WINO4) or cin % 8 or cout % 4:
        return False
    blocks = _lib.load().ud_conv3x3_wino4_f32_blocks(H, W)
    # (reductions longer than 1024 channels keep the 0.7 of rounds 3-4 on sparse maps: their F(4x4) error, 4.6e-5 at 2688
    # channels, is past the 2e-5 the small-map cases of tests/test_conv2d_f32_gpu.py allow)
    fill = WINO4_MIN_FILL if (cin <= 1024 or WINO4_MIN_FILL == 0.0) else max(WINO4_MIN_FILL, 0.7)
    if ((H + 3) // 4) * ((W + 3) // 4) < fill * 32 * blocks:
        return False
    return WINO4_MIN_FILL == 0.0 or cin >= WINO4_MIN_CIN or H * W >= 128 * 128


USE_WINO4_WGRAD = os.environ.get("UD_F32_WINO4_WGRAD", "1") != "0"
WINO4_WGRAD_ALL = False      # tests: every shape the kernel takes


def wino4_wgrad_pays(B, H, W, cin, cout):
    """Weight gradient through the F(4x4) form (ud_conv3x3_wino4_wgrad_nhwc_f32): same routing idea as the forward pass."""
    if not (USE_WINO4 and USE_WINO4_WGRAD) or cin % 32 or cout % 64 or B * H * W * max(cin, cout) * 4 >= 2 ** 31 - 1:
        return False
    # measured against the F(2x2) weight-gradient kernel (tools/time_wino4_wgrad.py): ahead on the large maps -- 64 -> 2688 @180^2
    # 1.87 -> 1.65 ms, 256 -> 128 0.40 -> 0.36, 64 -> 64 @64x176 x24 0.177 -> 0.138 -- level or behind on the small ones (its
    # transforms wait on a one-stage prefetch of dy; the F(2x2) kernel keeps two stages of raw tiles in registers)
    return WINO4_WGRAD_ALL or (wino4_pays(H, W, cin, cout) and H * W >= 11000)


def wino_pays(H, W, cin, cout):
    if not USE_WINOGRAD or cin % 8 or cout % 4:
        return False
    blocks = _lib.load().ud_conv3x3_wino_f32_blocks(H, W)
    return ((H + 1) // 2) * ((W + 1) // 2) >= WINO_MIN_FILL * 64 * blocks


def _wino_weights(weight, transposed, f4=False):
    """U = G g G^T in the kernel's stage order.  transposed: the data gradient's filter (Cin <-> Cout, taps reversed).  Cached on
    the tensor object for FROZEN weights only (requires_grad False: the distillation teacher), keyed by version + storage;
    trainable weights are transformed on every call -- fused optimizers update parameters without moving the version counter
    (measured: torch._fused_adamw_ leaves ``_version`` unchanged), so a version-keyed cache would serve stale filters."""
    lib = _lib.load()
    w = weight.detach()

    def make():
        n, c = (w.shape[1], w.shape[0]) if transposed else (w.shape[0], w.shape[1])
        sn, sc = (w.stride(1), w.stride(0)) if transposed else (w.stride(0), w.stride(1))
        nbytes, fn = ((lib.ud_conv3x3_wino4_f32_weight_bytes, lib.ud_conv3x3_wino4_f32_weights) if f4 else
                      (lib.ud_conv3x3_wino_f32_weight_bytes, lib.ud_conv3x3_wino_f32_weights))
        U = torch.empty(nbytes(c, n) // 4, dtype=torch.float32, device=w.device)
        _lib.check(fn(_lib.ptr(w), sn, sc, w.stride(2), w.stride(3), n, c, 1 if transposed else 0,
                      _lib.ptr(U), _lib.stream_of(w)), "ud_conv3x3_wino4_f32_weights" if f4 else "ud_conv3x3_wino_f32_weights")
        return U

    if weight.requires_grad or torch.cuda.is_current_stream_capturing():
        if weight.requires_grad and getattr(weight, "_ud_wino", None) is not None:
            weight._ud_wino = None          # a weight that is trained now and frozen again later must not find its old filters
        return make()
    key = (bool(transposed), bool(f4), weight._version, weight.data_ptr(), tuple(weight.shape), tuple(weight.stride()))
    cache = getattr(weight, "_ud_wino", None)
    if cache is not None and cache[0] == key:
        return cache[1]
    U = make()
    try:
        weight._ud_wino = (key, U)
    except (AttributeError, RuntimeError):
        pass
    return U


def _launch3(x, weight, bias=None, relu=False, transposed=False, bn_stats=False, scale=None, shift=None):
    """3x3 / stride 1 / pad 1 of channels-last x with the parameter ``weight`` [Cout, Cin, 3, 3] (transposed: its data-gradient
    filter, Cin <-> Cout with the taps reversed).  Winograd kernel where the map fills its tile blocks, else the direct one."""
    B, cin, H, W = x.shape
    cout = weight.shape[1] if transposed else weight.shape[0]
    if FLOP_COUNTER is not None:
        FLOP_COUNTER[0] += 2 * B * H * W * cout * 9 * cin
    if SHAPE_LOG is not None:
        SHAPE_LOG.append((B, cin, H, W, cout, bool(transposed)))
    lib = _lib.load()
    y = torch.empty((B, cout, H, W), dtype=torch.float32, device=x.device, memory_format=torch.channels_last)
    if wino4_pays(H, W, cin, cout):
        U = _wino_weights(weight, transposed, f4=True)
        part, ns = (None, ctypes.c_int(0))
        if bn_stats:
            part, ns = _bn_partial(lib.ud_conv3x3_wino4_bnstats_bytes(B, H, W, cout), x.device)
        ws = _lib.workspace(x.device, lib.ud_conv3x3_wino4_f32_workspace_bytes(B, H, W, cin, cout), "conv_w4")
        _lib.check(lib.ud_conv3x3_wino4_nhwc_f32(_lib.ptr(x), _lib.ptr(U), _lib.ptr(y), B, H, W, cin, cout, _lib.ptr(bias),
                                                 _lib.ptr(scale), _lib.ptr(shift), None, 1 if relu else 0, _lib.ptr(part),
                                                 part.numel() * 4 if bn_stats else 0, ctypes.addressof(ns), _lib.ptr(ws),
                                                 ws.numel(), _lib.stream_of(x)),
                   "ud_conv3x3_wino4_nhwc_f32")
        return (y, (part, ns.value, B * H * W)) if bn_stats else y
    if wino_pays(H, W, cin, cout):
        U = _wino_weights(weight, transposed)
        part, ns = (None, ctypes.c_int(0))
        if bn_stats:
            part, ns = _bn_partial(lib.ud_conv3x3_wino_bnstats_bytes(B, H, W, cout), x.device)
        _lib.check(lib.ud_conv3x3_wino_nhwc_f32(_lib.ptr(x), _lib.ptr(U), _lib.ptr(y), B, H, W, cin, cout, _lib.ptr(bias),
                                                _lib.ptr(scale), _lib.ptr(shift), None, 1 if relu else 0, _lib.ptr(part), part.numel() * 4 if bn_stats else 0,
                                                ctypes.addressof(ns), _lib.stream_of(x)), "ud_conv3x3_wino_nhwc_f32")
        return (y, (part, ns.value, B * H * W)) if bn_stats else y
    w = weight.detach()
    w_tap = (w.permute(1, 2, 3, 0) if transposed else w.permute(0, 2, 3, 1)).contiguous()   # [n, 3, 3, c]; transposed: taps walked in reverse
    if bn_stats:      # + per-tile (sum, sum of squares) for the BatchNorm that follows: (y, (partial, slices, rows))
        part, ns = _bn_partial(lib.ud_conv3x3_bnstats_bytes(B, H, W, cout), x.device)
        _lib.check(lib.ud_conv3x3_bnstats_nhwc_f32(_lib.ptr(x), _lib.ptr(w_tap), _lib.ptr(y), B, H, W, cin, cout,
                                                   _lib.ptr(bias), _lib.ptr(part), part.numel() * 4,
                                                   ctypes.addressof(ns), _lib.stream_of(x)),
                   "ud_conv3x3_bnstats_nhwc_f32")
        return y, (part, ns.value, B * H * W)
    _lib.check(lib.ud_conv3x3_nhwc_f32(_lib.ptr(x), _lib.ptr(w_tap), _lib.ptr(y), B, H, W, cin, cout,
                                       _lib.ptr(bias), _lib.ptr(scale), _lib.ptr(shift), None,
                                       (1 if relu else 0) | (2 if transposed else 0), _lib.stream_of(x)),
               "ud_conv3x3_nhwc_f32")
    return y


LOG_1X1 = None           # set to [] to record ("line", P, Cin, Cout) of every plain 1x1 launch (tools/time_f32_1x1.py)


# ud_conv1x1p_nhwc_f32 (csrc/conv2d_f32_1x1p.hip) takes the 1x1 launches it measured faster on, per shape of one distillation step
# (tools/time_1x1p.py): the persistent stream-K kernel from 12 slices of 32 channels on (16 896 x 1024 -> 256: 114 -> 89 us,
# 4 224 x 2048 -> 512: 142 -> 89 us; plain and mapped), its per-tile kernel with the epilogue from registers for the shorter plain
# reductions (16 896 x 256 -> 1024: 96 -> 90 us, 129 600 x 256 -> 128: 90 -> 75 us); 64-channel reductions stay on the
# grid-per-tile kernel of conv2d_f32.hip (270 336 x 64 -> 256: 116 vs 120 us: those launches are store-bound).
P1X1_MIN_K = int(os.environ.get("UD_F32_1X1P_MIN_K", "96"))
P1X1_MIN_K_MAPPED = int(os.environ.get("UD_F32_1X1P_MIN_K_MAPPED", "128"))
P1X1_STATS = os.environ.get("UD_F32_1X1P_STATS", "1") == "1"


# "1": mapped launches whose grid-per-tile grid is one nearly full round stay on that kernel (the rule of the per-slice staging;
# with the per-segment staging of the persistent kernel it no longer pays: 32 400 x 1152 -> 256 177 us against 182 us)
P1X1_KEEP_ONE_ROUND = os.environ.get("UD_F32_1X1P_ONE_ROUND", "0") == "1"


def persistent_1x1(K, mapped=False, P=0, N=0):
    if not (K >= (P1X1_MIN_K_MAPPED if mapped else P1X1_MIN_K) and _lib.load().ud_conv1x1_f32_persistent_enabled() == 1):
        return False
    if mapped and P and N and P1X1_KEEP_ONE_ROUND:
        # the grid-per-tile kernel keeps the mapped launches whose grid is ONE nearly full round of its 512 slots (the 90 x 90 x 4 BEV
        # maps: 254 tiles x 2 blocks of 128 channels = 508 workgroups; 32 400 x 1152 -> 256: 174 us against 215 us persistent --
        # tools/time_f32_1x1.py MAPPED=1), where its 128-wide tiles stage the mapped input half as often
        tiles = (P + 127) // 128
        wgs = tiles * ((N + 127) // 128)
        if N <= 64 or wgs <= 256:
            wgs = tiles * ((N + 63) // 64)
        if 460 <= wgs <= 512:
            return False
    return True


def launch_1x1p(x, w, y, P, K, N, bias=None, residual=None, part=None, imap=None, omap=None):
    """ud_conv1x1p_nhwc_f32 on torch tensors; -> number of BatchNorm partial rows (0 without `part`)."""
    lib = _lib.load()
    ws = _lib.workspace(x.device, lib.ud_conv1x1p_f32_workspace_bytes(), "conv_1x1p")
    ns = ctypes.c_int(0)
    _lib.check(lib.ud_conv1x1p_nhwc_f32(_lib.ptr(x), _lib.ptr(w), _lib.ptr(y), P, K, N, _lib.ptr(bias), None, None,
                                        _lib.ptr(residual), 0, _lib.ptr(part), part.numel() * 4 if part is not None else 0,
                                        ctypes.addressof(ns), imap, omap, x.numel(), y.numel(), _lib.ptr(ws), ws.numel(),
                                        _lib.stream_of(x)), "ud_conv1x1p_nhwc_f32")
    return ns.value


def _launch1(x, w, cout, bias=None, residual=None, bn_stats=False):
    B, cin, H, W = x.shape
    if LOG_1X1 is not None:
        LOG_1X1.append(("line", B * H * W, cin, cout, None, None))
    y = torch.empty((B, cout, H, W), dtype=torch.float32, device=x.device, memory_format=torch.channels_last)
    # BatchNorm-statistics launches included (P1X1_STATS): the register epilogue reduces its partial sums in DOUBLE (lane pairs, the 16-lane
    # exchange, the four waves) and rounds once -- with an fp32 lane tree the deblock BatchNorms of a randomly initialised image
    # branch turned the 1e-7 inconsistency between statistics and input into 2.8 % of a weight gradient
    # (tests/test_image_branch_f32_gpu.py, whose bounds stand as they were)
    if persistent_1x1(cin) and (P1X1_STATS or not bn_stats) and x.numel() < (1 << 30) - (1 << 18) \
            and y.numel() < (1 << 30) - (1 << 18):
        if bn_stats:
            part, _ = _bn_partial(_lib.load().ud_conv1x1_bnstats_bytes(B * H * W, cout), x.device)
            ns = launch_1x1p(x, w, y, B * H * W, cin, cout, bias=bias, part=part)
            return y, (part, ns, B * H * W)
        launch_1x1p(x, w, y, B * H * W, cin, cout, bias=bias, residual=residual)
        return y
    if bn_stats:
        lib = _lib.load()
        part, ns = _bn_partial(lib.ud_conv1x1_bnstats_bytes(B * H * W, cout), x.device)
        _lib.check(lib.ud_conv1x1_bnstats_nhwc_f32(_lib.ptr(x), _lib.ptr(w), _lib.ptr(y), B * H * W, cin, cout,
                                                   _lib.ptr(bias), _lib.ptr(part), part.numel() * 4,
                                                   ctypes.addressof(ns), _lib.stream_of(x)),
                   "ud_conv1x1_bnstats_nhwc_f32")
        return y, (part, ns.value, B * H * W)
    _lib.check(_lib.load().ud_conv1x1_nhwc_f32(_lib.ptr(x), _lib.ptr(w), _lib.ptr(y), B * H * W, cin, cout,
                                               _lib.ptr(bias), None, None, _lib.ptr(residual), 0, _lib.stream_of(x)),
               "ud_conv1x1_nhwc_f32")
    return y


USE_HIP_WGRAD = True       # False: aten.convolution_backward (MIOpen fp32 split-K kernels, atomics) for A/B timing


def wgrad_supported(x, gy):
    return (x.is_cuda and x.dtype == torch.float32 and gy.dtype == torch.float32 and x.shape[1] % 4 == 0
            and gy.shape[1] % 4 == 0)


def weight_grad(x, gy, w, ks):
    """dW of a stride-1 'same' convolution; x, gy channels-last fp32, w [Cout, Cin, ks, ks].  Hand-written fp32 MFMA
    kernels with a fixed-order slice reduction (ud_conv3x3_wgrad_nhwc_f32 / ud_conv1x1_wgrad_mapped_nhwc_f32); the
    result comes back in the parameter's channels-last strides (a [Cout, ks, ks, Cin] buffer viewed as [Cout, Cin, ks, ks])."""
    cout, cin = w.shape[0], w.shape[1]
    if not (USE_HIP_WGRAD and wgrad_supported(x, gy)):
        if USE_HIP_WGRAD:
            _lib.library_fallthrough("ops.conv2d_f32.weight_grad", x, gy, w, kernel=ks)
        p = ks // 2
        return torch.ops.aten.convolution_backward(gy, x, w, None, [1, 1], [p, p], [1, 1], False, [0, 0], 1,
                                                   [False, True, False])[1]
    lib = _lib.load()
    B, _, H, W = x.shape
    if ks == 3:
        wino = USE_WINOGRAD and USE_WINOGRAD_WGRAD
        if wino and wino4_wgrad_pays(B, H, W, cin, cout):
            ws = _lib.workspace(x.device, lib.ud_conv3x3_wino4_wgrad_f32_workspace_bytes(B, H, W, cin, cout), "conv_wgrad")
            dw = torch.empty((cout, 3, 3, cin), dtype=torch.float32, device=x.device)
            _lib.check(lib.ud_conv3x3_wino4_wgrad_nhwc_f32(_lib.ptr(x), _lib.ptr(gy), _lib.ptr(dw), B, H, W, cin, cout, _lib.ptr(ws),
                                                           ws.numel(), _lib.stream_of(x)), "ud_conv3x3_wino4_wgrad_nhwc_f32")
            return dw.permute(0, 3, 1, 2)
        nbytes, fn = ((lib.ud_conv3x3_wino_wgrad_f32_workspace_bytes, lib.ud_conv3x3_wino_wgrad_nhwc_f32) if wino else
                      (lib.ud_conv3x3_wgrad_f32_workspace_bytes, lib.ud_conv3x3_wgrad_nhwc_f32))
        ws = _lib.workspace(x.device, nbytes(B, H, W, cin, cout), "conv_wgrad")
        dw = torch.empty((cout, 3, 3, cin), dtype=torch.float32, device=x.device)
        _lib.check(fn(_lib.ptr(x), _lib.ptr(gy), _lib.ptr(dw), B, H, W, cin, cout, _lib.ptr(ws), ws.numel(), _lib.stream_of(x)),
                   "ud_conv3x3_wino_wgrad_nhwc_f32" if wino else "ud_conv3x3_wgrad_nhwc_f32")
        return dw.permute(0, 3, 1, 2)
    return wgrad_mapped(x, gy, B * H * W, cin, cout, None, None).view(cout, cin, 1, 1)


def wgrad_mapped(x, gy, P, K, N, xmap, ymap):
    """dW [N][K] = sum_p dy'[p][N] x'[p][K] with pixel maps (ops/conv2d.py:_pmap) on x and / or dy; fp32."""
    lib = _lib.load()
    ws = _lib.workspace(x.device, lib.ud_conv1x1_wgrad_f32_workspace_bytes(P, K, N), "conv_wgrad")
    dw = torch.empty((N, K), dtype=torch.float32, device=x.device)
    _lib.check(lib.ud_conv1x1_wgrad_mapped_nhwc_f32(_lib.ptr(x), _lib.ptr(gy), _lib.ptr(dw), P, K, N, xmap, ymap,
                                                    _lib.ptr(ws), ws.numel(), _lib.stream_of(x)),
               "ud_conv1x1_wgrad_mapped_nhwc_f32")
    return dw


class _ConvF32(torch.autograd.Function):
    """with_skip (1x1 only): the function also returns its input (an alias) for the caller's identity branch, so that the
    gradient arriving through that branch is added in the data-gradient kernel's epilogue instead of by a separate
    elementwise add of two full-size tensors (ResNet residual joins; same contract as ops/conv2d.py:_Conv1x1Fn)."""

    @staticmethod
    def forward(ctx, x, weight, bias, ks, with_skip=False, holder=None):
        _lib.require_gpu(x, weight)
        x = _nhwc(x)
        b = None if bias is None else bias.detach().contiguous()
        w = weight.detach()
        st = holder is not None
        if ks == 3:
            y = _launch3(x, weight, b, bn_stats=st)
        else:
            y = _launch1(x, w.reshape(weight.shape[0], weight.shape[1]).contiguous(), weight.shape[0], b, bn_stats=st)
        if st:
            y, part = y
            holder.append(part)
        ctx.save_for_backward(x, weight)
        ctx.ks, ctx.has_bias = ks, bias is not None
        return (y, x) if with_skip else y

    @staticmethod
    def backward(ctx, gy, gskip=None):
        x, weight = ctx.saved_tensors
        ks = ctx.ks
        gy = _nhwc(gy.float())
        if gskip is not None:
            gskip = _nhwc(gskip.float())
        gx = gw = gb = None
        w = weight.detach()
        p = ks // 2
        if ctx.needs_input_grad[0] and weight.shape[0] % 32 != 0:
            # the data gradient reduces over Cout in 32-channel slices: zero-pad dy and the transposed weights to the next
            # multiple (depth net 512 -> 368: one 25 MB copy) instead of leaving the hand-written path
            cout, pad = weight.shape[0], (-weight.shape[0]) % 32
            gyp = torch.empty((gy.shape[0], cout + pad, gy.shape[2], gy.shape[3]), dtype=gy.dtype, device=gy.device,
                              memory_format=torch.channels_last)
            gyp[:, :cout] = gy
            gyp[:, cout:] = 0
            if ks == 3:
                wp = torch.zeros((cout + pad,) + tuple(w.shape[1:]), dtype=w.dtype, device=w.device)
                wp[:cout] = w
                gx = _launch3(gyp, wp, transposed=True)
                if gskip is not None:
                    gx = gx + gskip
            else:
                wt = torch.zeros((weight.shape[1], cout + pad), dtype=w.dtype, device=w.device)
                wt[:, :cout] = w.reshape(cout, weight.shape[1]).t()
                gx = _launch1(gyp, wt, weight.shape[1], residual=gskip)
        elif ctx.needs_input_grad[0]:
            if ks == 3:      # un-flipped transposed weights [Cin, 3, 3, Cout], taps walked in reverse
                gx = _launch3(gy, weight, transposed=True)
            else:
                gx = _launch1(gy, w.reshape(weight.shape[0], weight.shape[1]).t().contiguous(), weight.shape[1],
                              residual=gskip)
        if ctx.needs_input_grad[1]:
            gw = wgrad_stream.defer(weight, lambda: weight_grad(x, gy, w, ks), x, gy)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            gb = bn_act.bias_grad(gy)
        return gx, gw, gb, None, None, None


def _apply(x, weight, bias, ks, with_skip, bn_stats):
    """bn_stats: a training-mode BatchNorm follows -- its statistics pass is folded into the convolution's epilogue and the
    partials ride on the output tensor (ops.bn_act.bn_act picks them up)."""
    if not bn_stats:
        return _ConvF32.apply(x, weight, bias, ks, with_skip)
    holder = []
    y = _ConvF32.apply(x, weight, bias, ks, with_skip, holder)
    if holder:
        (y[0] if with_skip else y)._ud_bn_partial = holder[0]
    return y


def conv3x3(x, weight, bias=None, bn_stats=False):
    return _apply(x, weight, bias, 3, False, bn_stats)


def conv3x3_inference(x, weight, bias=None, scale=None, shift=None, relu=False):
    """Inference 3x3 convolution with the fused epilogue (+ bias) (* scale + shift: a folded eval-mode BatchNorm) (ReLU): the frozen
    teacher's conv -> BN -> ReLU links in one kernel (no autograd graph)."""
    with torch.no_grad():
        _lib.require_gpu(x, weight)
        b = None if bias is None else bias.detach().float().contiguous()
        return _launch3(_nhwc(x), weight, b, relu=relu, scale=scale, shift=shift)


def conv1x1(x, weight, bias=None, bn_stats=False):
    return _apply(x, weight, bias, 1, False, bn_stats)


def conv1x1_skip(x, weight, bias=None, bn_stats=False):
    """(conv1x1(x), x'): x' aliases x; a gradient arriving through x' is added inside the data-gradient kernel."""
    return _apply(x, weight, bias, 1, True, bn_stats)
