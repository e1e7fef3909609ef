"""GPU: weight gradients on their own stream (ops/wgrad_stream.py) are bit-identical to the inline computation, arrive in the
parameter's strides (AccumulateGrad takes them over without a copy), are joined when the backward pass ends -- also under
torch.autograd.grad -- and fall back to the inline path when the parameter already holds a gradient (accumulation, DDP bucket views)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _cl(t):
    return t.contiguous(memory_format=torch.channels_last)


def _layers():
    from unidistill_amd.ops import conv2d as c, conv2d_f32 as c32
    return [("conv3x3", lambda x, w: c32.conv3x3(x, w), (64, 64, 3, 3), (2, 64, 20, 36)),
            ("conv1x1", lambda x, w: c32.conv1x1(x, w), (128, 64, 1, 1), (2, 64, 20, 36)),
            ("patch", lambda x, w: c.conv_patch(x, w, 2), (128, 64, 2, 2), (2, 64, 20, 36)),
            ("tpatch", lambda x, w: c.conv_transpose_patch(x, w, 2), (64, 128, 2, 2), (2, 64, 10, 18)),
            ("s3x3", lambda x, w: c.conv3x3_stride2(x, w), (128, 64, 3, 3), (2, 64, 20, 36))]


@pytest.mark.parametrize("use_backward", [True, False])
def test_side_stream_weight_gradients_equal_inline(hip_lib, monkeypatch, use_backward):
    from unidistill_amd.ops import wgrad_stream as ws
    for name, f, wshape, xshape in _layers():
        torch.manual_seed(len(name))
        x = _cl(torch.randn(*xshape, device="cuda"))
        w0 = torch.randn(*wshape, device="cuda") * 0.05
        if wshape[2] > 1:
            w0 = _cl(w0)
        res = []
        for on in (True, False):
            monkeypatch.setattr(ws, "ENABLED", on)
            w = torch.nn.Parameter(w0.clone(memory_format=torch.preserve_format))
            calls = []
            real = torch.Tensor.record_stream
            monkeypatch.setattr(torch.Tensor, "record_stream", lambda t, s: (calls.append(1), real(t, s))[1])
            y = f(x, w)
            gy = torch.randn(y.shape, device="cuda", generator=torch.Generator("cuda").manual_seed(7)).contiguous(memory_format=torch.channels_last)
            if use_backward:
                y.backward(gy)
                g = w.grad
            else:
                g, = torch.autograd.grad(y, (w,), gy)
            monkeypatch.setattr(torch.Tensor, "record_stream", real)
            assert bool(calls) == on, f"{name}: deferral {'did not engage' if on else 'engaged while switched off'}"
            assert (not on) or g.stride() == w.stride(), f"{name}: gradient strides {g.stride()} != parameter strides {w.stride()}"
            res.append(g.clone())          # read on the current stream right away: the engine callback must have joined the side stream
        assert torch.equal(res[0], res[1]), name


def test_parameter_with_a_gradient_takes_the_inline_path(hip_lib, monkeypatch):
    from unidistill_amd.ops import conv2d_f32 as c32, wgrad_stream as ws
    monkeypatch.setattr(ws, "ENABLED", True)
    x = _cl(torch.randn(2, 64, 12, 20, device="cuda"))
    w = torch.nn.Parameter(torch.randn(64, 64, 1, 1, device="cuda") * 0.05)
    c32.conv1x1(x, w).sum().backward()
    first = w.grad.clone()
    calls = []
    real = torch.Tensor.record_stream
    monkeypatch.setattr(torch.Tensor, "record_stream", lambda t, s: (calls.append(1), real(t, s))[1])
    c32.conv1x1(x, w).sum().backward()              # accumulation: AccumulateGrad adds on the caller's stream
    assert not calls and torch.equal(w.grad, 2 * first)


def test_parameter_used_twice_in_one_backward(hip_lib, monkeypatch):
    """A module applied twice in one graph (or tied weights): autograd's input buffer adds the two weight gradients on the caller's
    stream BEFORE AccumulateGrad -- the second use must wait for the side stream and compute inline.  The first gradient is made
    slow on purpose (a large map) and the second tiny, so that an unsynchronised add would read the first one half-written."""
    from unidistill_amd.ops import conv2d_f32 as c32, wgrad_stream as ws
    torch.manual_seed(11)
    w0 = _cl(torch.randn(64, 64, 3, 3, device="cuda") * 0.05)
    xa = _cl(torch.randn(4, 64, 180, 180, device="cuda"))
    xb = _cl(torch.randn(1, 64, 8, 8, device="cuda"))
    res = []
    for on in (True, False):
        monkeypatch.setattr(ws, "ENABLED", on)
        before = dict(ws.STATS)
        for rep in range(3):
            w = torch.nn.Parameter(w0.clone(memory_format=torch.preserve_format))
            # graph order: the big convolution's backward runs first (it is the later node), then the small one's
            loss = c32.conv3x3(xb, w).square().sum() + c32.conv3x3(xa, w).square().sum()
            loss.backward()
            res.append(w.grad.clone())
        if on:
            assert ws.STATS["inline_repeat"] - before["inline_repeat"] == 3 and ws.STATS["deferred"] - before["deferred"] == 3
    for g in res[1:]:
        assert torch.equal(g, res[0])


def test_backward_that_died_leaves_no_unsynchronised_gradients(hip_lib, monkeypatch):
    """A backward pass that raises after a deferral never reaches its engine callback; the next pass must join the leftovers and
    queue a callback of its own."""
    from unidistill_amd.ops import conv2d_f32 as c32, wgrad_stream as ws
    monkeypatch.setattr(ws, "ENABLED", True)
    x = _cl(torch.randn(2, 64, 40, 40, device="cuda"))
    w = torch.nn.Parameter(_cl(torch.randn(64, 64, 3, 3, device="cuda") * 0.05))

    class Boom(torch.autograd.Function):
        @staticmethod
        def forward(ctx, t):
            return t.clone()

        @staticmethod
        def backward(ctx, g):
            raise RuntimeError("boom")
    y = c32.conv3x3(Boom.apply(x.requires_grad_(True)), w)
    with pytest.raises(RuntimeError, match="boom"):
        y.sum().backward()                      # the convolution's backward (deferral) runs before Boom's
    assert ws._pending                          # left behind by the dead pass
    w2 = torch.nn.Parameter(_cl(torch.randn(64, 64, 3, 3, device="cuda") * 0.05))
    xin = _cl(torch.randn(2, 64, 40, 40, device="cuda"))
    c32.conv3x3(xin, w2).sum().backward()
    assert not ws._pending and not ws._deferred     # joined by this pass's own callback
    monkeypatch.setattr(ws, "ENABLED", False)
    w3 = torch.nn.Parameter(w2.detach().clone(memory_format=torch.preserve_format))
    c32.conv3x3(xin, w3).sum().backward()
    assert torch.equal(w2.grad, w3.grad)
