"""Weight gradients on their own HIP stream.

In backward a convolution's weight gradient depends only on (x, dy) and nothing but gradient accumulation / the optimizer reads it,
so the MFMA-bound weight-gradient kernels can run beside the HBM-bound BatchNorm / elementwise kernels of the main stream's chain
(the reference runs a layer's three backward kernels back to back: base_bev_backbone.py:38-115, mmdet ResNet).  Measured on the
fp32 distillation step: 60.1 -> 58.6 ms.

``defer(weight, thunk, x, dy)`` is called from a convolution's backward:
  * the side stream waits for the point the caller's stream has reached (x and dy are ready there), runs ``thunk`` with its own
    scratch buffers, and materialises the result in EXACTLY the parameter's strides -- autograd's AccumulateGrad then takes the
    tensor over without launching anything (a layout mismatch would make it copy on the caller's stream, unsynchronised);
  * x / dy / the result are recorded on the streams that use them (caching-allocator reuse);
  * one callback queued on the autograd engine makes the caller's stream wait for the side stream when the backward pass ends,
    whoever started it (Trainer.step, torch.autograd.grad in a test).
It computes inline -- the behaviour of rounds 1-5 -- when the parameter already has a gradient (accumulation steps:
AccumulateGrad then adds on the caller's stream) or carries a hook, under create_graph, on CPU, with ``UD_WGRAD_STREAM=0``, and
under DistributedDataParallel (``disable()``, called by train.Trainer: DDP's reducer hook copies every gradient into its bucket
on the caller's stream as soon as AccumulateGrad has run, i.e. before the join).
(An engine-level variant -- an identity node recorded on the side stream so that autograd orders the streams itself -- was built
first and gave non-reproducible losses; it was not pursued.)"""
import os

import torch
from torch.autograd import Variable

from .. import _lib

ENABLED = os.environ.get("UD_WGRAD_STREAM", "1") == "1"
_streams = {}
_pending = set()


def join():
    """The current stream waits for the weight gradients enqueued so far (runs by itself when a backward pass ends; Trainer.step
    calls it once more so that a backward that died half-way cannot leave a later one unsynchronised)."""
    for idx in list(_pending):
        torch.cuda.current_stream(idx).wait_stream(_streams[idx])
    _pending.clear()


def disable():
    """Inline weight gradients for the rest of the process (DistributedDataParallel: see the module text)."""
    global ENABLED
    ENABLED = False


def defer(weight, thunk, *keep):
    """dW of ``weight`` = thunk(), computed on the weight-gradient stream when that is safe (see the module text)."""
    if ENABLED and torch.distributed.is_available() and torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1:
        disable()           # a process group with peers: assume a gradient reducer (DDP) hangs on the parameters
    if not (ENABLED and weight.is_cuda and weight.is_leaf and weight.grad is None and not torch.is_grad_enabled()
            and not weight._backward_hooks and not getattr(weight, "_post_accumulate_grad_hooks", None)
            and not torch.cuda.is_current_stream_capturing()):
        return thunk()      # (a hook on the parameter would read the gradient on the caller's stream)
    idx = weight.device.index
    side = _streams.get(idx)
    if side is None:
        side = _streams[idx] = torch.cuda.Stream(weight.device)
    cur = torch.cuda.current_stream(idx)
    side.wait_stream(cur)
    with torch.cuda.stream(side), _lib.workspace_scope("wgrad_stream"):
        g = thunk()
        if g.dtype != weight.dtype or g.shape != weight.shape or g.stride() != weight.stride():
            out = torch.empty_like(weight)              # the parameter's own strides (preserve_format)
            out.copy_(g)
            g = out
    for t in keep:
        t.record_stream(side)                           # their blocks must not be handed out again before this stream has read them
    g.record_stream(cur)                                # allocated in the side stream's pool, read by the optimizer on the caller's
    first = not _pending
    _pending.add(idx)
    if first:
        try:
            Variable._execution_engine.queue_callback(join)
        except RuntimeError:                            # not inside an autograd backward pass (a Function's backward called by hand)
            join()
    return g
