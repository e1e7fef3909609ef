"""CPU: the autograd plumbing of the fused concatenation (ops/bn_act.cat_buffer / cat_slices; reference base_bev_backbone.py:117-141,
`torch.cat(ups, dim=1)`): producers write channel slices of ONE channels-last buffer, the concatenated map is that buffer, and each
producer's backward receives its slice of the gradient as a strided view.  The HIP kernels that use it are tested in
tests/test_bn_act_gpu.py; here a plain torch producer stands in for them."""
import torch

from unidistill_amd.ops import bn_act as hb


class _Twice(torch.autograd.Function):
    seen = []

    @staticmethod
    def forward(ctx, x, out):
        out[0].copy_(x * 2)
        return out[0]

    @staticmethod
    def backward(ctx, g):
        _Twice.seen.append((tuple(g.shape), tuple(g.stride())))
        return g * 2, None


def test_cat_slices_equals_torch_cat_and_hands_over_strided_gradient_slices():
    torch.manual_seed(0)
    B, H, W, widths = 2, 5, 7, [16, 8, 24]
    xs = [torch.randn(B, c, H, W, requires_grad=True) for c in widths]
    buf, slots = hb.cat_buffer(torch.empty(B, 1, H, W), widths)
    assert buf.is_contiguous(memory_format=torch.channels_last) and buf.shape == (B, sum(widths), H, W)
    assert [s.shape[1] for s in slots] == widths and all(not s._is_view() for s in slots)
    _Twice.seen.clear()
    parts = [_Twice.apply(x, (s,)) for x, s in zip(xs, slots)]
    y = hb.cat_slices(buf, parts)
    ref = torch.cat([x * 2 for x in xs], 1)
    assert torch.equal(y, ref) and y.data_ptr() == buf.data_ptr()
    w = torch.randn_like(y)
    (y * w).sum().backward()
    c0 = 0
    for x, c in zip(xs, widths):
        assert torch.equal(x.grad, 2 * w[:, c0:c0 + c])
        c0 += c
    ct = sum(widths)
    assert sorted(_Twice.seen) == sorted(((B, c, H, W), (H * W * ct, 1, W * ct, ct)) for c in widths)      # no copies: row stride = the wide map's


def test_row_stride_recognises_channel_slices_only():
    ref = torch.empty(2, 16, 5, 7).contiguous(memory_format=torch.channels_last)
    if not torch.cuda.is_available():
        # _row_stride answers for device tensors only (the kernels' inputs); on CPU it declines
        assert hb._row_stride(ref, ref) is None
        return
    wide = torch.empty(2, 48, 5, 7, device="cuda").contiguous(memory_format=torch.channels_last)
    r = ref.cuda()
    assert hb._row_stride(r, r) == 16
    assert hb._row_stride(wide[:, 16:32], r) == 48
    assert hb._row_stride(torch.empty(2, 16, 5, 7, device="cuda"), r) is None        # NCHW strides: not rows of channels
