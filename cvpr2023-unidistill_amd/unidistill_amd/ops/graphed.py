"""hipGraph capture of a shape-static training segment: forward AND backward of a module as two graphs.

A distillation step enqueues ~1 490 kernels from Python (22-25 ms of CPU time per step on the threads that enqueue it, bench.py
`host_enqueue_ms`); ~60 % of them belong to the camera branch -- ResNet-50 + SECONDFPN + depth net over 24 images of 256 x 704
(the reference builds it through mmdet / mmdet3d: layers/blocks_3d/mmdet3d/lss_fpn.py:143-149,242-262) -- whose shapes never
change.  ``GraphedModule`` captures that segment once per (input shape, dtype, autocast state) with
``torch.cuda.make_graphed_callables`` (forward graph + backward graph, replayed from an autograd Function) and replays it from
then on: two launches instead of ~900, the same kernels in the same order -- results are bit-identical to the eager path
(tests/test_graphed_gpu.py).

What capture needs from the ops of this package, and how they provide it:
  * no host reads and no allocation-dependent branches inside the segment (the image branch has none);
  * scratch buffers that nobody else touches: the capture runs inside ``_lib.workspace_scope("graph:<name>")``, whose buffers are
    sized by the warm-up iterations and never re-allocated (``_lib.workspace`` raises during capture otherwise);
  * weight gradients: ``ops.wgrad_stream.defer`` forks the weight-gradient stream INTO the capture (side.wait_stream) and the
    engine callback that ends the captured backward joins it again, so the graph keeps the overlap of the eager step
    (UD_GRAPH_SIDE_STREAM=0: computed inline on the capturing stream);
  * BatchNorm running statistics / num_batches_tracked are updated by kernels, i.e. by the REPLAY; the three warm-up iterations
    and the capture itself would advance them four times too often, so the module's buffers are restored after capture.
Anything else (eval mode, no_grad, a new shape) runs the wrapped module eagerly.
"""
import torch
from torch import nn

from .. import _lib


import os

SIDE_STREAM_IN_CAPTURE = os.environ.get("UD_GRAPH_SIDE_STREAM", "1") == "1"


class _Segment(nn.Module):
    """Children of a parent module as one callable with parameters(), WITHOUT entering the parent's module tree (the parent's
    state_dict keeps the reference's key names)."""

    def __init__(self, fn, modules):
        super().__init__()
        self.parts = nn.ModuleList(modules)
        self._fn = fn

    def forward(self, x):
        return self._fn(x)


class GraphedModule:
    def __init__(self, name, fn, modules):
        self.name = name
        self.segment = _Segment(fn, modules)
        self.graphed = {}
        self.enabled = True
        self.replays = 0

    def _key(self, x):
        ac = torch.get_autocast_dtype("cuda") if torch.is_autocast_enabled() else None
        return (tuple(x.shape), tuple(x.stride()), x.dtype, ac)

    def _capture(self, x):
        seg = self.segment
        bufs = [b for b in seg.buffers()]
        saved = [b.detach().clone() for b in bufs]
        sample = x.detach().clone()
        from . import wgrad_stream
        # warm-up sizes the SAME workspaces the capture bakes in: the weight-gradient stream takes part in both (it joins the capture:
        # fork at every deferral, join in the engine callback that ends the captured backward) with scratch buffers of this graph's own
        was = (wgrad_stream.CAPTURE_OK, wgrad_stream.SCOPE[0], wgrad_stream.ENABLED)
        wgrad_stream.CAPTURE_OK, wgrad_stream.SCOPE[0] = SIDE_STREAM_IN_CAPTURE, "graph:" + self.name + ":wgrad"
        if not SIDE_STREAM_IN_CAPTURE:
            wgrad_stream.ENABLED = False
        try:
            g = self._make(seg, sample)
        finally:
            wgrad_stream.CAPTURE_OK, wgrad_stream.SCOPE[0], wgrad_stream.ENABLED = was
        with torch.no_grad():
            torch._foreach_copy_(bufs, saved)          # warm-up + capture advanced the running statistics: put them back
        return g           # == seg, whose instance-level forward now replays the graphs (the eager path calls seg._fn directly)

    def _make(self, seg, sample):
        with _lib.workspace_scope("graph:" + self.name), torch.autocast("cuda", enabled=torch.is_autocast_enabled(),
                                                                        dtype=torch.get_autocast_dtype("cuda"),
                                                                        cache_enabled=False):
            return torch.cuda.make_graphed_callables(seg, (sample,), num_warmup_iters=3, allow_unused_input=True)

    def __call__(self, x):
        seg = self.segment
        if not (self.enabled and x.is_cuda and torch.is_grad_enabled() and all(m.training for m in seg.parts)
                and any(p.requires_grad for p in seg.parameters())):
            return seg._fn(x)
        key = self._key(x)
        g = self.graphed.get(key)
        if g is None:
            g = self.graphed[key] = self._capture(x)
        self.replays += 1
        return g(x)        # a torch.autograd.Function that copies x into the static input and replays the forward graph
