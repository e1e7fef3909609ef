"""Sparse 3-D convolution: the slice of spconv's python API the reference uses, on HIP kernels.

Reference call sites: unidistill/layers/blocks_3d/det3d/spconv_backbone.py:3-5 (imports),
:21-48 (SubMConv3d / SparseConv3d / SparseInverseConv3d kwargs), :354-359 (SparseConvTensor),
:61-113 (replace_feature, .features), height_compression.py:19 (.dense()).

Differences by design (MI355X-first, see csrc/spconv_index.hip):
  * every site set owns a rank-bitmap index; rulebooks are dense [sites, K] neighbour tables and
    are cached per SITE SET and conv geometry, so two layers with different ``indice_key`` but the
    same sites/geometry ("subm1" / "res1") share one rulebook;
  * strided-conv outputs come out in ascending (b, z, y, x) order, deterministically;
  * conv / dgrad / wgrad are output-stationary fp32-MFMA kernels without scatter-add atomics.
Weights use spconv 2.x's KRSC layout [out, kz, ky, kx, in].
"""
import ctypes
import math

import torch
from torch import nn

from .. import _lib
from . import bn_act


def _triple(v):
    if isinstance(v, (list, tuple)):
        assert len(v) == 3
        return tuple(int(x) for x in v)
    return (int(v),) * 3


def _i3(v):
    return (ctypes.c_int * 3)(*v)


class _SiteSet:
    """Active sites of one resolution level + lazily built index and rulebooks."""

    def __init__(self, indices, spatial_shape, batch_size, rows_sorted):
        self.indices = indices                      # i32[M,4] (b,z,y,x), contiguous, cuda
        self.spatial_shape = tuple(int(s) for s in spatial_shape)
        self.batch_size = int(batch_size)
        self.rows_sorted = bool(rows_sorted)
        self._index = None
        self._subm = {}
        self._down = {}

    @property
    def M(self):
        return self.indices.shape[0]

    def _grid(self):
        return (self.batch_size,) + self.spatial_shape

    def index(self):
        if self._index is None:
            lib = _lib.load()
            B, Dz, Hy, Wx = self._grid()
            nbytes = lib.ud_spconv_index_bytes(B, Dz, Hy, Wx, self.M)
            if nbytes == 0:
                raise ValueError(f"invalid sparse grid {self._grid()}")
            buf = torch.empty(nbytes, dtype=torch.uint8, device=self.indices.device)
            _lib.check(lib.ud_spconv_build_index(_lib.ptr(self.indices), self.M, B, Dz, Hy, Wx,
                                                 1 if self.rows_sorted else 0, _lib.ptr(buf), nbytes,
                                                 _lib.stream_of(buf)), "ud_spconv_build_index")
            self._index = buf
        return self._index

    def subm_rulebook(self, ksize):
        rb = self._subm.get(ksize)
        if rb is None:
            B, Dz, Hy, Wx = self._grid()
            K = ksize[0] * ksize[1] * ksize[2]
            rb = torch.empty((self.M, K), dtype=torch.int32, device=self.indices.device)
            _lib.check(_lib.load().ud_spconv_subm_rulebook(
                _lib.ptr(self.index()), 1 if self.rows_sorted else 0, _lib.ptr(self.indices), self.M,
                B, Dz, Hy, Wx, ksize[0], ksize[1], ksize[2], _lib.ptr(rb), _lib.stream_of(rb)),
                "ud_spconv_subm_rulebook")
            self._subm[ksize] = rb
        return rb

    def down(self, ksize, stride, pad):
        """-> (output _SiteSet, out_nbr i32[Mout,K], in_nbr i32[Min,K])."""
        key = (ksize, stride, pad)
        hit = self._down.get(key)
        if hit is not None:
            return hit
        lib = _lib.load()
        dev = self.indices.device
        B, Dz, Hy, Wx = self._grid()
        out_shape = tuple((d + 2 * p - k) // s + 1 for d, k, s, p in
                          zip(self.spatial_shape, ksize, stride, pad))
        K = ksize[0] * ksize[1] * ksize[2]
        reach = 1
        for k, s in zip(ksize, stride):
            reach *= (k + s - 1) // s             # outputs one input can touch per axis
        cells = B * out_shape[0] * out_shape[1] * out_shape[2]
        cap = max(min(self.M * reach, cells), 1)
        nbytes = lib.ud_spconv_index_bytes(B, *out_shape, cap)
        out_index = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        out_coords = torch.empty((cap, 4), dtype=torch.int32, device=dev)
        m_out = torch.empty(1, dtype=torch.int32, device=dev)
        st = _lib.stream_of(out_coords)
        _lib.check(lib.ud_spconv_down_outputs(_lib.ptr(self.indices), self.M, B, Dz, Hy, Wx,
                                              _i3(ksize), _i3(stride), _i3(pad), _lib.ptr(out_index),
                                              nbytes, _lib.ptr(out_coords), cap, _lib.ptr(m_out), st),
                   "ud_spconv_down_outputs")
        Mout = int(m_out.item())                  # one host read per new level (sizes the tensors)
        out_coords = out_coords[:Mout]
        out_set = _SiteSet(out_coords, out_shape, B, rows_sorted=True)
        out_set._index = out_index
        out_nbr = torch.empty((Mout, K), dtype=torch.int32, device=dev)
        in_nbr = torch.empty((self.M, K), dtype=torch.int32, device=dev)
        _lib.check(lib.ud_spconv_down_rulebook(_lib.ptr(self.index()), 1 if self.rows_sorted else 0,
                                               self.M, B, Dz, Hy, Wx, _i3(ksize), _i3(stride),
                                               _i3(pad), _lib.ptr(out_coords), Mout,
                                               _lib.ptr(out_nbr), _lib.ptr(in_nbr), st),
                   "ud_spconv_down_rulebook")
        hit = (out_set, out_nbr, in_nbr)
        self._down[key] = hit
        return hit


class DeferredPyramid:
    """Site sets of a strided-conv chain built WITHOUT host reads in between: every level's output-site kernels take their
    input row count from the device word the previous level (or the voxelizer) wrote, on launches that cover an upper
    bound.  ``counts()`` is the tensor to read (once, together with anything else); ``finalize(host_counts)`` slices the
    cap-sized buffers, wires the _SiteSet chain and builds the rulebooks with the exact sizes.
    (spconv's API -- and round 1-2 here -- reads one size per level: five blocking reads per encoder pass.)"""

    def __init__(self, coords_cap, m_dev, spatial_shape, batch_size, geoms):
        """coords_cap i32[cap,4]; m_dev: 1-element int32 device tensor (rows in use); geoms: [(ksize, stride, pad)]."""
        lib = _lib.load()
        dev = coords_cap.device
        st = _lib.stream_of(coords_cap)
        self.coords_cap, self.shape0, self.B, self.geoms = coords_cap, tuple(int(v) for v in spatial_shape), int(batch_size), geoms
        cap = coords_cap.shape[0]
        B = self.B
        nbytes = lib.ud_spconv_index_bytes(B, *self.shape0, cap)
        self.index0 = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        _lib.check(lib.ud_spconv_build_index_dev(_lib.ptr(coords_cap), _lib.ptr(m_dev), cap, B, *self.shape0, 0,
                                                 _lib.ptr(self.index0), nbytes, st), "ud_spconv_build_index_dev")
        self.levels = []
        in_coords, in_cnt, in_cap, in_shape = coords_cap, m_dev, cap, self.shape0
        for ksize, stride, pad in geoms:
            out_shape = tuple((d + 2 * p - k) // s + 1 for d, k, s, p in zip(in_shape, ksize, stride, pad))
            reach = 1
            for k, s_ in zip(ksize, stride):
                reach *= (k + s_ - 1) // s_
            cells = B * out_shape[0] * out_shape[1] * out_shape[2]
            ocap = max(min(in_cap * reach, cells), 1)
            ob = lib.ud_spconv_index_bytes(B, *out_shape, ocap)
            out_index = torch.empty(ob, dtype=torch.uint8, device=dev)
            out_coords = torch.empty((ocap, 4), dtype=torch.int32, device=dev)
            m_out = torch.empty(1, dtype=torch.int32, device=dev)
            _lib.check(lib.ud_spconv_down_outputs_dev(_lib.ptr(in_coords), _lib.ptr(in_cnt), in_cap, B, *in_shape,
                                                      _i3(ksize), _i3(stride), _i3(pad), _lib.ptr(out_index), ob,
                                                      _lib.ptr(out_coords), ocap, _lib.ptr(m_out), st),
                       "ud_spconv_down_outputs_dev")
            self.levels.append((out_shape, out_index, out_coords, m_out))
            in_coords, in_cnt, in_cap, in_shape = out_coords, m_out, ocap, out_shape

    def counts(self):
        return [lv[3] for lv in self.levels]

    def finalize(self, M, level_counts):
        """M rows at level 0, level_counts[i] rows at strided level i (host ints) -> the level-0 _SiteSet, chain wired."""
        lib = _lib.load()
        root = _SiteSet(self.coords_cap[:M], self.shape0, self.B, rows_sorted=False)
        root._index = self.index0
        cur = root
        for (ksize, stride, pad), (out_shape, out_index, out_coords, _), Mout in zip(self.geoms, self.levels, level_counts):
            Mout = int(Mout)
            K = ksize[0] * ksize[1] * ksize[2]
            dev = out_coords.device
            oc = out_coords[:Mout]
            out_set = _SiteSet(oc, out_shape, self.B, rows_sorted=True)
            out_set._index = out_index
            out_nbr = torch.empty((Mout, K), dtype=torch.int32, device=dev)
            in_nbr = torch.empty((cur.M, K), dtype=torch.int32, device=dev)
            B, Dz, Hy, Wx = cur._grid()
            _lib.check(lib.ud_spconv_down_rulebook(_lib.ptr(cur.index()), 1 if cur.rows_sorted else 0, cur.M, B, Dz, Hy, Wx,
                                                   _i3(ksize), _i3(stride), _i3(pad), _lib.ptr(oc), Mout,
                                                   _lib.ptr(out_nbr), _lib.ptr(in_nbr), _lib.stream_of(oc)),
                       "ud_spconv_down_rulebook")
            cur._down[(ksize, stride, pad)] = (out_set, out_nbr, in_nbr)
            cur = out_set
        return root


_ORDER_CACHE_ATTR = "_ud_mask_order"


def mask_order(nbr, mirror=False):
    """Permutation of the rulebook rows sorted by their neighbour bit mask (cached on the rulebook
    tensor).  Rows with equal / similar masks become neighbours, so a 128-row tile activates only a
    few of the K kernel offsets instead of nearly all of them."""
    # ``mirror`` only permutes the bits of every mask: rows with equal masks are neighbours either way,
    # so the forward order also serves the mirrored (submanifold data-gradient) pass.
    key = _ORDER_CACHE_ATTR
    order = getattr(nbr, key, None)
    if order is None:
        K = nbr.shape[1]
        if K > 31 or nbr.shape[0] == 0:
            order = False
        else:
            # one C-ABI call (ud_spconv_mask_order): offset frequencies from a row sample, bit weights by rarity -- the
            # most frequent offset (the centre, the in-plane faces) gets bit 0, the rarest (the corners) the top bits, so
            # rows group by their RARE neighbours first (20.2 vs 20.6 of 27 active offsets per tile at the 128-channel
            # level, 0.915 vs 0.902 useful MFMA rows per wave) -- masks, and a stable radix sort over the K mask bits.
            # Ascending = tiles with the most active offsets last; the kernels dispatch the LAST tile first.
            lib = _lib.load()
            M = nbr.shape[0]
            ws = _lib.workspace(nbr.device, lib.ud_spconv_mask_order_workspace_bytes(M, K), "mask_order")
            order = torch.empty(M, dtype=torch.int32, device=nbr.device)
            _lib.check(lib.ud_spconv_mask_order(_lib.ptr(nbr), M, K, _lib.ptr(order), _lib.ptr(ws), ws.numel(),
                                                _lib.stream_of(nbr)), "ud_spconv_mask_order")
        setattr(nbr, key, order)
    return None if order is False else order


CONV_LOG = None          # set to [] to record (rulebook, Cin, Cout, "f32" | "bf16") of every conv launch (bench.py)


def _conv(feat, nbr, weight, w_strides, mirror, bias, cin, cout, algo=0, scale=None, shift=None,
          residual=None, relu=False):
    Mout, K = nbr.shape
    if CONV_LOG is not None:
        CONV_LOG.append((nbr, cin, cout, "f32"))
    out = torch.empty((Mout, cout), dtype=torch.float32, device=feat.device)
    order = mask_order(nbr, mirror) if algo in (0, 3) else None
    if feat.numel() * feat.element_size() >= 0xFFFF0000:
        raise ValueError("ud_spconv_conv: the input feature tensor must be smaller than 4 GiB - 64 KiB (32-bit byte offsets)")
    _lib.check(_lib.load().ud_spconv_conv(_lib.ptr(feat), _lib.ptr(nbr), _lib.ptr(weight),
                                          w_strides[0], w_strides[1], w_strides[2],
                                          1 if mirror else 0, _lib.ptr(bias), _lib.ptr(out), Mout, K,
                                          cin, cout, algo, _lib.ptr(order), _lib.ptr(scale),
                                          _lib.ptr(shift), _lib.ptr(residual), 1 if relu else 0,
                                          _lib.stream_of(feat)), "ud_spconv_conv")
    return out


def tile_masks(nbr):
    """u32 activity mask per 64-row tile of ``nbr`` in its mask-sorted row order; cached on the rulebook
    tensor like the row order itself (both depend on the rulebook only)."""
    hit = getattr(nbr, "_ud_tile_masks", None)
    if hit is None:
        Mout, K = nbr.shape
        hit = torch.empty(((Mout + 63) // 64,), dtype=torch.int32, device=nbr.device)
        _lib.check(_lib.load().ud_spconv_tile_masks(_lib.ptr(nbr), Mout, K, _lib.ptr(mask_order(nbr, False)),
                                                    _lib.ptr(hit), _lib.stream_of(nbr)), "ud_spconv_tile_masks")
        nbr._ud_tile_masks = hit
    return hit


def sorted_rulebook(nbr):
    """nbr rows in mask-sorted order (cached on the rulebook): the weight-gradient kernel then walks
    contiguous tiles of the (equally re-ordered) output gradient instead of chasing ``order``."""
    hit = getattr(nbr, "_ud_sorted", None)
    if hit is None:
        order = mask_order(nbr, False)
        hit = nbr if order is None else nbr.index_select(0, order.long()).contiguous()
        nbr._ud_sorted = hit
    return hit


def _conv_bf16io(feat, nbr, weight, bias, cin, cout, scale=None, shift=None, residual=None, relu=False,
                 mirror=False):
    """Conv with bf16 tensors in HBM (ud_spconv_conv_bf16io): feat fp32 or bf16 [*, cin], weight
    [cout, K, cin] bf16 (or fp32 when cin % 4 != 0), output (and residual) bf16 [Mout, cout].
    ``mirror`` reads rulebook column K-1-k for weight offset k (submanifold data gradient)."""
    Mout, K = nbr.shape
    if CONV_LOG is not None:
        CONV_LOG.append((nbr, cin, cout, "bf16"))
    io = 2 | (4 if weight.dtype == torch.bfloat16 else 0)
    if feat.dtype == torch.bfloat16 and cin % 4 == 0:
        io |= 1
    elif feat.dtype != torch.float32:
        feat = feat.float()
    if residual is not None and residual.dtype != torch.bfloat16:
        residual = residual.to(torch.bfloat16)
    out = torch.empty((Mout, cout), dtype=torch.bfloat16, device=feat.device)
    order = mask_order(nbr, mirror)
    _lib.check(_lib.load().ud_spconv_conv_bf16io(_lib.ptr(feat), _lib.ptr(nbr), _lib.ptr(weight),
                                                 K * cin, cin, 1, 1 if mirror else 0, _lib.ptr(bias),
                                                 _lib.ptr(out),
                                                 Mout, K, cin, cout, io, _lib.ptr(order),
                                                 _lib.ptr(scale), _lib.ptr(shift), _lib.ptr(residual),
                                                 1 if relu else 0, _lib.stream_of(feat)),
               "ud_spconv_conv_bf16io")
    return out


def bf16_weight(conv):
    """[cout, K, cin] bf16 copy of a sparse conv's weight for the mixed-precision inference path;
    cached on the module, refreshed when the parameter changes (version counter)."""
    w = conv.weight
    ver = (w._version, w.device, w.data_ptr())
    hit = getattr(conv, "_ud_w_bf16", None)
    if hit is None or hit[0] != ver or w.requires_grad:
        with torch.no_grad():
            hit = (ver, w.detach().reshape(w.shape[0], -1, w.shape[-1]).to(torch.bfloat16).contiguous())
        conv._ud_w_bf16 = hit
    return hit[1]


def folded_batchnorm(bn):
    """(scale, shift) of an eval-mode BatchNorm1d: y = x * scale + shift.  Cached on the module
    and refreshed when any of its tensors changed (version counters)."""
    ver = (bn.weight._version, bn.bias._version, bn.running_mean._version, bn.running_var._version,
           bn.weight.device)
    hit = getattr(bn, "_ud_folded", None)
    # trainable parameters are never cached: fused optimizers step them without a version bump
    if hit is None or hit[0] != ver or bn.weight.requires_grad or bn.bias.requires_grad:
        with torch.no_grad():
            scale = (bn.weight / torch.sqrt(bn.running_var + bn.eps)).float().contiguous()
            shift = (bn.bias - bn.running_mean * scale).float().contiguous()
        hit = (ver, scale, shift)
        bn._ud_folded = hit
    return hit[1], hit[2]


def wants_grad(*objs):
    """True when autograd is recording and any of the tensors / any parameter of the modules needs a
    gradient -- then an inference-only fused kernel (which builds no graph) must NOT be used: with a
    frozen BatchNorm in eval mode but trainable conv weights the input alone says nothing."""
    if not torch.is_grad_enabled():
        return False
    for o in objs:
        if o is None:
            continue
        if torch.is_tensor(o):
            if o.requires_grad:
                return True
        elif any(p.requires_grad for p in o.parameters()):
            return True
    return False


def can_fuse_inference(x, bn=None, *modules):
    """The fused conv(+BN+residual+ReLU) epilogue is inference-only: BN in eval and nobody -- the input,
    the conv(s) or the BatchNorm(s) in ``modules`` -- needs a gradient."""
    if wants_grad(x.features, bn, *modules):
        return False
    return bn is None or (isinstance(bn, nn.BatchNorm1d) and not bn.training and bn.affine
                          and bn.track_running_stats)


def effective_algo(algo):
    """Kernel choice for this call: under bf16 autocast the default MFMA path (0) becomes the
    bf16-operand / fp32-accumulate kernel (3); explicit choices (1, 2, 3) are kept."""
    if algo == 0 and torch.is_autocast_enabled("cuda") and torch.get_autocast_dtype("cuda") == torch.bfloat16:
        return 3
    return algo


class _SparseConvFn(torch.autograd.Function):
    """out = conv(features; nbr, W, bias).  nbr_t / mirror describe the transposed rulebook.

    fp32 mode: exact fp32 MFMA kernels, fp32 tensors.  Under bf16 autocast (algo 3) the activations
    are bf16 in HBM end to end -- forward and data gradient on the all-bf16 kernel, weight gradient on
    the bf16 transposing-read kernel -- with fp32 accumulation and fp32 master weights."""

    @staticmethod
    def forward(ctx, features, weight, bias, nbr, nbr_t, mirror_t, algo):
        _lib.require_gpu(features, weight, nbr)
        cout, cin = weight.shape[0], weight.shape[-1]
        K = nbr.shape[1]
        assert weight.numel() == cout * K * cin and features.shape[1] == cin
        algo = effective_algo(algo)
        b32 = None if bias is None else bias.detach().contiguous().float()
        if algo == 3 and K <= 32 and cin % 8 == 0 and cout % 8 == 0:
            xb = features.detach().to(torch.bfloat16).contiguous()
            wb = weight.detach().reshape(cout, K, cin).to(torch.bfloat16).contiguous()
            out = _conv_bf16io(xb, nbr, wb, b32, cin, cout)
            ctx.save_for_backward(xb, wb, nbr, nbr_t)
            ctx.cfg = (mirror_t, "bf16io", bias is not None, weight.shape, features.dtype)
            return out
        features = features.contiguous().float()
        w = weight.contiguous().float()
        out = _conv(features, nbr, w, (K * cin, cin, 1), False, b32, cin, cout, algo)
        ctx.save_for_backward(features, w, nbr, nbr_t)
        ctx.cfg = (mirror_t, algo, bias is not None, weight.shape, torch.float32)
        return out

    @staticmethod
    def backward(ctx, gout):
        features, w, nbr, nbr_t = ctx.saved_tensors
        mirror_t, algo, has_bias, wshape, in_dtype = ctx.cfg
        cout, cin = wshape[0], wshape[-1]
        K = nbr.shape[1]
        lib = _lib.load()
        Mout = nbr.shape[0]
        gin = gw = gb = None
        if algo == "bf16io":
            gout = gout.to(torch.bfloat16).contiguous()
            if ctx.needs_input_grad[0]:
                # transposed conv: reduce over cout with W^T laid out [cin, K, cout]
                wt = w.permute(2, 1, 0).contiguous()
                gin = _conv_bf16io(gout, nbr_t, wt, None, cout, cin, mirror=mirror_t).to(in_dtype)
            if ctx.needs_input_grad[1]:
                gw = torch.empty(wshape, dtype=torch.float32, device=w.device)
                need = lib.ud_spconv_wgrad_bf16_workspace_bytes(Mout, K, cin, cout)
                ws = _lib.workspace(w.device, need, "spconv_wgrad")
                order = mask_order(nbr, False)
                if order is not None and cin in (32, 64, 128) and cout in (32, 64, 128):
                    # rulebook pre-sorted (cached), gout rows located through row_order inside the kernel
                    g_rows, io, row_order = gout, 3, order
                    nbr_rows = sorted_rulebook(nbr)
                else:
                    # narrow layers (register-staged kernel): it follows row_order for both operands itself
                    g_rows, io, row_order, nbr_rows = gout, 1, order, nbr
                _lib.check(lib.ud_spconv_wgrad_bf16(_lib.ptr(features), _lib.ptr(nbr_rows),
                                                    _lib.ptr(g_rows), _lib.ptr(gw), Mout, K, cin, cout, io,
                                                    _lib.ptr(row_order), _lib.ptr(tile_masks(nbr)), _lib.ptr(ws),
                                                    ws.numel(), _lib.stream_of(w)), "ud_spconv_wgrad_bf16")
            if has_bias and ctx.needs_input_grad[2]:
                gb = bn_act.bias_grad(gout)
            return gin, gw, gb, None, None, None, None
        gout = gout.contiguous().float()
        if ctx.needs_input_grad[0]:
            # transposed conv: reduce over cout; W element (n'=c, k, c'=n) at c + k*cin + n*K*cin
            gin = _conv(gout, nbr_t, w, (1, cin, K * cin), mirror_t, None, cout, cin, algo)
        if ctx.needs_input_grad[1]:
            gw = torch.empty(wshape, dtype=torch.float32, device=w.device)
            if algo == 3 and K <= 32:       # bf16 operands staged from fp32 tensors
                need = lib.ud_spconv_wgrad_bf16_workspace_bytes(Mout, K, cin, cout)
                ws = _lib.workspace(w.device, need, "spconv_wgrad")
                _lib.check(lib.ud_spconv_wgrad_bf16(_lib.ptr(features), _lib.ptr(nbr), _lib.ptr(gout),
                                                    _lib.ptr(gw), Mout, K, cin, cout, 0,
                                                    _lib.ptr(mask_order(nbr, False)),
                                                    _lib.ptr(tile_masks(nbr)), _lib.ptr(ws),
                                                    ws.numel(), _lib.stream_of(w)), "ud_spconv_wgrad_bf16")
            else:
                need = lib.ud_spconv_wgrad_workspace_bytes(Mout, K, cin, cout)
                ws = _lib.workspace(w.device, need, "spconv_wgrad")
                _lib.check(lib.ud_spconv_wgrad(_lib.ptr(features), _lib.ptr(nbr), _lib.ptr(gout),
                                               _lib.ptr(gw), Mout, K, cin, cout, algo, _lib.ptr(ws),
                                               ws.numel(), _lib.stream_of(w)), "ud_spconv_wgrad")
        if has_bias and ctx.needs_input_grad[2]:
            gb = bn_act.bias_grad(gout)
        return gin, gw, gb, None, None, None, None


class _DenseFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, features, indices, grid):
        features = features.contiguous().float()
        B, Dz, Hy, Wx = grid
        M, C = features.shape
        dense = torch.empty((B, C, Dz, Hy, Wx), dtype=torch.float32, device=features.device)
        lib = _lib.load()
        ws = _lib.workspace(features.device, lib.ud_sparse_bev_workspace_bytes(B, Dz, Hy, Wx), "sparse_bev")
        _lib.check(lib.ud_sparse_to_dense(_lib.ptr(features), _lib.ptr(indices), M, C, B, Dz, Hy, Wx,
                                          _lib.ptr(dense), _lib.ptr(ws), ws.numel(), _lib.stream_of(dense)),
                   "ud_sparse_to_dense")
        ctx.save_for_backward(indices)
        ctx.grid = grid
        ctx.mc = (M, C)
        return dense

    @staticmethod
    def backward(ctx, gdense):
        (indices,) = ctx.saved_tensors
        B, Dz, Hy, Wx = ctx.grid
        M, C = ctx.mc
        gdense = gdense.contiguous().float()
        g = torch.zeros((M, C), dtype=torch.float32, device=gdense.device)    # rows outside the grid: 0
        lib = _lib.load()
        ws = _lib.workspace(g.device, lib.ud_sparse_bev_workspace_bytes(B, Dz, Hy, Wx), "sparse_bev")
        _lib.check(lib.ud_dense_to_sparse(_lib.ptr(gdense), _lib.ptr(indices), M, C, B, Dz, Hy, Wx,
                                          _lib.ptr(g), _lib.ptr(ws), ws.numel(), _lib.stream_of(g)),
                   "ud_dense_to_sparse")
        return g, None, None


FUSED_BEV = True         # False: bev() = dense().view(...) (tests, A/B timing)


class _BevFn(torch.autograd.Function):
    """features [M, C] (bf16 or fp32) -> channels-last BEV map [B, C*Dz, Hy, Wx] of the same dtype in one pass
    (ud_sparse_to_bev_bf16 / ud_sparse_to_bev_f32) -- HeightCompression without the NCDHW tensor and its layout copy."""

    @staticmethod
    def forward(ctx, features, indices, grid):
        lib = _lib.load()
        features = features.contiguous()
        B, Dz, Hy, Wx = grid
        M, C = features.shape
        f32 = features.dtype == torch.float32
        bev = torch.empty((B, Hy, Wx, C * Dz), dtype=features.dtype, device=features.device)
        ws = _lib.workspace(features.device, lib.ud_sparse_bev_workspace_bytes(B, Dz, Hy, Wx), "sparse_bev")
        fn = lib.ud_sparse_to_bev_f32 if f32 else lib.ud_sparse_to_bev_bf16
        _lib.check(fn(_lib.ptr(features), _lib.ptr(indices), M, C, B, Dz, Hy, Wx, _lib.ptr(bev), _lib.ptr(ws), ws.numel(),
                      _lib.stream_of(bev)), "ud_sparse_to_bev")
        ctx.save_for_backward(indices)
        ctx.cfg = (grid, M, C, features.dtype)
        return bev.permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, gbev):
        (indices,) = ctx.saved_tensors
        (B, Dz, Hy, Wx), M, C, dt = ctx.cfg
        lib = _lib.load()
        g = gbev.to(dt).permute(0, 2, 3, 1).contiguous()
        gfeat = torch.zeros((M, C), dtype=dt, device=g.device)
        ws = _lib.workspace(g.device, lib.ud_sparse_bev_workspace_bytes(B, Dz, Hy, Wx), "sparse_bev")
        fn = lib.ud_bev_to_sparse_f32 if dt == torch.float32 else lib.ud_bev_to_sparse_bf16
        _lib.check(fn(_lib.ptr(g), _lib.ptr(indices), M, C, B, Dz, Hy, Wx, _lib.ptr(gfeat), _lib.ptr(ws), ws.numel(),
                      _lib.stream_of(g)), "ud_bev_to_sparse")
        return gfeat, None, None


class SparseConvTensor:
    """features f32[M,C] + indices i32[M,4] (b,z,y,x) on a (spatial_shape, batch_size) grid."""

    def __init__(self, features, indices, spatial_shape, batch_size, _sites=None, _indice_dict=None):
        _lib.require_gpu(features, indices)
        self.features = features
        if _sites is None:
            idx = indices if indices.dtype == torch.int32 else indices.int()
            _sites = _SiteSet(idx.contiguous(), spatial_shape, batch_size, rows_sorted=False)
        self._sites = _sites
        self.indice_dict = {} if _indice_dict is None else _indice_dict

    @property
    def indices(self):
        return self._sites.indices

    @property
    def spatial_shape(self):
        return list(self._sites.spatial_shape)

    @property
    def batch_size(self):
        return self._sites.batch_size

    def replace_feature(self, feature):
        return SparseConvTensor(feature, None, None, None, _sites=self._sites,
                                _indice_dict=self.indice_dict)

    def bev(self):
        """[B, C*Dz, Hy, Wx]: z folded into channels (channel = c*Dz + z) -- HeightCompression.  bf16
        features give a channels-last bf16 map in one kernel; otherwise dense().view(...)."""
        grid = (self.batch_size,) + tuple(self._sites.spatial_shape)
        C = self.features.shape[1]
        if self.features.dtype in (torch.bfloat16, torch.float32) and (C * grid[1]) % 4 == 0 and self.indices.shape[0] > 0 \
                and FUSED_BEV:
            return _BevFn.apply(self.features, self.indices, grid)
        d = self.dense()
        n, c, dz, h, w = d.shape
        return d.view(n, c * dz, h, w)

    def dense(self, channels_first=True):
        grid = (self.batch_size,) + tuple(self._sites.spatial_shape)
        feats = self.features if self.features.dtype == torch.float32 else self.features.float()
        out = _DenseFn.apply(feats, self.indices, grid)
        return out if channels_first else out.permute(0, 2, 3, 4, 1).contiguous()


class SparseModule(nn.Module):
    """Marker base: modules that take and return SparseConvTensor."""


class SparseSequential(SparseModule):
    """nn.Sequential that routes dense layers (BatchNorm1d, ReLU, ...) over ``.features``."""

    def __init__(self, *mods):
        super().__init__()
        for i, m in enumerate(mods):
            self.add_module(str(i), m)

    def __len__(self):
        return len(self._modules)

    def __getitem__(self, i):
        return list(self._modules.values())[i]

    def forward(self, x):
        mods = list(self._modules.values())
        i = 0
        while i < len(mods):
            m = mods[i]
            if isinstance(m, _SparseConvBase) and isinstance(x, SparseConvTensor) and m.kernel_algo == 0:
                # inference fast path: conv [+ BatchNorm1d(eval)] [+ ReLU] as one kernel
                bn = mods[i + 1] if i + 1 < len(mods) and isinstance(mods[i + 1], nn.BatchNorm1d) else None
                if x.indices.shape[0] != 0 and can_fuse_inference(x, bn, m) and (bn is None or not bn.training):
                    j = i + (2 if bn is not None else 1)
                    relu = j < len(mods) and isinstance(mods[j], nn.ReLU)
                    if bn is not None or relu:
                        x = m.forward_fused(x, bn, relu)
                        i = j + (1 if relu else 0)
                        continue
            if isinstance(m, SparseModule):
                x = m(x)
            elif isinstance(x, SparseConvTensor):
                if x.indices.shape[0] != 0:
                    if isinstance(m, nn.BatchNorm1d):
                        # BatchNorm1d (+ ReLU) over the voxel rows: streaming HIP kernels in bf16 mode
                        from ..layers.dense import batchnorm_act
                        relu = i + 1 < len(mods) and isinstance(mods[i + 1], nn.ReLU)
                        x = x.replace_feature(batchnorm_act(m, x.features, None, relu))
                        i += 2 if relu else 1
                        continue
                    x = x.replace_feature(m(x.features))
            else:
                x = m(x)
            i += 1
        return x


class _SparseConvBase(SparseModule):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1,
                 groups=1, bias=True, subm=False, inverse=False, indice_key=None, algo=None, **_):
        super().__init__()
        assert groups == 1 and _triple(dilation) == (1, 1, 1), "groups/dilation are not used by UniDistill"
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size = _triple(kernel_size)
        self.stride = _triple(stride)
        self.padding = _triple(padding)
        self.subm, self.inverse = subm, inverse
        self.indice_key = indice_key
        self.algo = algo
        self.kernel_algo = 0     # 0 = MFMA where instantiated, 1 = generic VALU (tests)
        self.weight = nn.Parameter(torch.empty(out_channels, *self.kernel_size, in_channels))
        self.bias = nn.Parameter(torch.empty(out_channels)) if bias else None
        self.reset_parameters()

    def reset_parameters(self):
        fan_in = self.in_channels * self.kernel_size[0] * self.kernel_size[1] * self.kernel_size[2]
        bound = 1.0 / math.sqrt(fan_in)
        with torch.no_grad():
            self.weight.uniform_(-bound, bound)   # == kaiming_uniform_(a=sqrt(5))
            if self.bias is not None:
                self.bias.uniform_(-bound, bound)

    def _rulebooks(self, x):
        """-> (out_sites, nbr, nbr_t, mirror_t) for this conv on tensor x (cached per site set)."""
        sites = x._sites
        if self.subm:
            nbr = sites.subm_rulebook(self.kernel_size)
            return sites, nbr, nbr, True
        if self.inverse:
            src = x.indice_dict.get(self.indice_key)
            if src is None:
                raise ValueError(f"SparseInverseConv3d needs the SparseConv3d with indice_key "
                                 f"{self.indice_key!r} to have run first")
            in_sites, out_nbr, in_nbr = src
            assert sites is in_sites[1], "inverse conv input must be that conv's output"
            return in_sites[0], in_nbr, out_nbr, False
        out_sites, nbr, nbr_t = sites.down(self.kernel_size, self.stride, self.padding)
        if self.indice_key is not None:
            x.indice_dict[self.indice_key] = ((sites, out_sites), nbr, nbr_t)
        return out_sites, nbr, nbr_t, False

    def forward(self, x):
        assert isinstance(x, SparseConvTensor)
        out_sites, nbr, nbr_t, mirror = self._rulebooks(x)
        feats = _SparseConvFn.apply(x.features, self.weight, self.bias, nbr, nbr_t, mirror,
                                    self.kernel_algo)
        return SparseConvTensor(feats, None, None, None, _sites=out_sites, _indice_dict=x.indice_dict)

    def forward_fused(self, x, bn=None, relu=False, residual=None):
        """Inference only: conv + bias (+ folded BatchNorm1d) (+ residual) (+ ReLU) in ONE kernel."""
        out_sites, nbr, _, _ = self._rulebooks(x)
        w = self.weight.detach().contiguous().float()
        cout, cin = w.shape[0], w.shape[-1]
        K = nbr.shape[1]
        scale = shift = None
        if bn is not None:
            scale, shift = folded_batchnorm(bn)
        with torch.no_grad():
            bias = None if self.bias is None else self.bias.detach().contiguous().float()
            if effective_algo(0) == 3 and K <= 32:
                # bf16 autocast: activations stay bf16 between the fused layers (half the gather bytes)
                feats = _conv_bf16io(x.features.detach().contiguous(), nbr,
                                     bf16_weight(self) if cin % 4 == 0 else w.view(cout, K, cin), bias, cin,
                                     cout, scale, shift,
                                     None if residual is None else residual.detach().contiguous(), relu)
            else:
                feats = _conv(x.features.detach().contiguous().float(), nbr, w, (K * cin, cin, 1), False,
                              bias, cin, cout, 0, scale, shift,
                              None if residual is None else residual.detach().contiguous().float(), relu)
        return SparseConvTensor(feats, None, None, None, _sites=out_sites, _indice_dict=x.indice_dict)


class SubMConv3d(_SparseConvBase):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1,
                 groups=1, bias=True, indice_key=None, algo=None, **kw):
        super().__init__(in_channels, out_channels, kernel_size, stride, padding, dilation, groups,
                         bias, subm=True, indice_key=indice_key, algo=algo)


class SparseConv3d(_SparseConvBase):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1,
                 groups=1, bias=True, indice_key=None, algo=None, **kw):
        super().__init__(in_channels, out_channels, kernel_size, stride, padding, dilation, groups,
                         bias, subm=False, indice_key=indice_key, algo=algo)


class SparseInverseConv3d(_SparseConvBase):
    def __init__(self, in_channels, out_channels, kernel_size, indice_key=None, bias=True,
                 algo=None, **kw):
        super().__init__(in_channels, out_channels, kernel_size, 1, 0, 1, 1, bias, subm=False,
                         inverse=True, indice_key=indice_key, algo=algo)


class ConvAlgo:
    """spconv.core.ConvAlgo stand-in; the value is accepted and ignored (one algorithm here)."""
    Native = 0
    MaskImplicitGemm = 1
    MaskSplitImplicitGemm = 2
