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
  * one callback queued on the autograd engine PER BACKWARD PASS (graph task) makes the caller's stream wait for the side stream
    when the pass ends, whoever started it (Trainer.step, torch.autograd.grad in a test); entries a pass that died half-way left
    behind are joined by the next pass's first ``defer``.
It computes inline -- the behaviour of rounds 1-4 -- when the parameter already has a gradient (accumulation steps: AccumulateGrad
then adds on the caller's stream) or carries a hook that is not known to be stream-safe, under create_graph, on CPU, with
``UD_WGRAD_STREAM=0``, and for the SECOND use of a parameter inside one backward pass (a module called twice, tied weights):
autograd's input buffer adds the two gradients on the caller's stream before AccumulateGrad, so the caller's stream first waits for
the side stream and the second gradient is computed inline (tests/test_wgrad_stream_gpu.py).  Not covered: a parameter that is ALSO
used by an op outside this package in the same graph (its gradient never passes through ``defer``).

**Under DistributedDataParallel** (the reference's only launch mode: exps/base_cli.py:40-45 ``accelerator="ddp"``) the reducer's
hook reads every gradient on the caller's stream right after AccumulateGrad -- it copies it into the bucket -- i.e. before any join.
``attach_ddp(ddp)`` (train.Trainer) therefore registers a communication hook and hands the reducer gradients it does not have
to touch:
  * the hook records every parameter's BUCKET VIEW (``GradBucket.gradients()``); once the views are the same on two consecutive
    steps (DDP re-buckets exactly once, before its second forward), ``defer`` makes the side stream write dW straight into a fresh
    alias of that view and returns it: the reducer finds ``grad.is_alias_of(bucket_view)`` and launches nothing (it also no longer
    copies ~300 gradients per step into the buckets, which zero_grad(set_to_none=True) otherwise costs);
  * before a bucket's all-reduce is enqueued the hook makes the caller's stream wait for the side stream (three buckets per step:
    three waits instead of one per layer), then runs the stock ``allreduce_hook`` (division by the world size + async all-reduce,
    which orders the communication stream behind the caller's stream);
  * if a bucket's buffer ever moves (a re-bucketing this module did not expect) every recorded view is dropped and the steps fall
    back to inline gradients until the views are stable again.
A process group with peers but NO attached reducer (somebody else's gradient hook) keeps the inline path.
(An engine-level variant -- an identity node recorded on the side stream so that autograd orders the streams itself -- was built
first and gave non-reproducible losses; it was not pursued.)"""
import os

import torch
from torch.autograd import Variable

from .. import _lib

ENABLED = os.environ.get("UD_WGRAD_STREAM", "1") == "1"
_streams = {}
_pending = set()            # device indices with weight gradients enqueued since the last join
_deferred = set()           # storage addresses of the parameters deferred since the last join
_task = [None]              # the autograd graph task whose completion callback is queued
_ddp = {"on": False, "views": {}, "buffers": {}}
STATS = {"deferred": 0, "inline_repeat": 0, "ddp_direct": 0, "ddp_inline": 0}

_DEBUG = os.environ.get("UD_WGRAD_DEBUG")
# hipGraph capture (ops/graphed.py): the side stream may JOIN a capture -- the fork (side.wait_stream) and the join (the engine
# callback at the end of the captured backward) both land inside it -- with scratch buffers of the capture's own (SCOPE): the
# captured weight gradients run beside whatever the eager weight-gradient stream is doing at replay time
CAPTURE_OK = False
SCOPE = ["wgrad_stream"]

_graph_task_id = getattr(torch._C, "_current_graph_task_id", None)


def _current_task():
    return _graph_task_id() if _graph_task_id is not None else -1


def join():
    """The current stream waits for the weight gradients enqueued so far (runs by itself when a backward pass ends; Trainer.step
    calls it once more so that a backward that died half-way cannot leave a later one unsynchronised)."""
    for idx in list(_pending):
        torch.cuda.current_stream(idx).wait_stream(_streams[idx])
    _pending.clear()
    _deferred.clear()
    _task[0] = None


def disable():
    """Inline weight gradients for the rest of the process."""
    global ENABLED
    ENABLED = False


def state():
    """'on' | 'off' | 'ddp' -- for the bench line (config.wgrad_stream)."""
    return "off" if not ENABLED else ("ddp" if _ddp["on"] else "on")


# ---- DistributedDataParallel ---------------------------------------------------------------------------------------------------

def bucket_ready(bucket):
    """Communication-hook half: called when DDP is about to reduce ``bucket`` (autograd thread, caller's stream current)."""
    buf = bucket.buffer()
    if not buf.is_cuda:
        return
    idx = buf.device.index
    side = _streams.get(idx)
    if side is not None and idx in _pending:
        torch.cuda.current_stream(idx).wait_stream(side)       # every dW written into this bucket so far is complete behind this point
    bi = bucket.index()
    known = _ddp["buffers"].get(bi)
    here = (buf.data_ptr(), buf.numel() * buf.element_size())
    if known is not None and known[0] == here and known[1] >= 2:
        return                                                  # the views of this bucket are recorded and have been stable
    if known is not None and known[0] != here:
        if STATS["ddp_direct"]:
            # gradients were already being written into the recorded views: this step's went into the OLD buffers and the reducer
            # copied them on the caller's stream without waiting for the side stream -- never continue silently
            raise RuntimeError("wgrad_stream: DistributedDataParallel moved a gradient bucket after its views had been stable "
                               "for two steps (re-bucketing is expected once, before the second forward); set UD_WGRAD_STREAM=0")
        _ddp["views"].clear()                                   # the bucket moved: nothing recorded can be trusted
        _ddp["buffers"].clear()
        known = None
    views = _ddp["views"]
    same = known is not None
    why = None
    for p, g in zip(bucket.parameters(), bucket.gradients()):
        key = p.data_ptr()
        # GradBucket.gradients() hands out CONTIGUOUS slices of the flat buffer; the reducer's own views (and therefore the
        # gradients it accepts without a copy) carry the parameter's strides when the parameter is dense (reducer.cpp
        # initialize_bucket_views: as_strided(sizes, strides, offset) over the same numel elements): rebuild exactly that
        if g.stride() != p.stride():
            if not _dense(p):
                views.pop(key, None)
                continue
            g = g.as_strided(p.shape, p.stride(), g.storage_offset())
        prev = views.get(key)
        if prev is None or prev.data_ptr() != g.data_ptr() or prev.stride() != g.stride() or prev.shape != g.shape:
            if same or why is None:
                why = ("new" if prev is None else "moved", tuple(p.shape), tuple(g.shape), tuple(g.stride()))
            same = False
        views[key] = g
    _ddp["buffers"][bi] = (here, (known[1] + 1) if same else 1)
    if _DEBUG:
        with open(_DEBUG + ".rank%d" % torch.distributed.get_rank(), "a") as f:
            f.write("bucket %d ptr %x bytes %d params %d count %d first mismatch %s stats %s\n" % (
                bi, here[0], here[1], len(bucket.parameters()), _ddp["buffers"][bi][1], why, STATS))


def _dense(t):
    """Non-overlapping and dense: some permutation of a contiguous layout (channels-last filters)."""
    expect = 1
    for n, st in sorted(((n, st) for n, st in zip(t.shape, t.stride()) if n > 1), key=lambda v: v[1]):
        if st != expect:
            return False
        expect *= n
    return True


def _ddp_view(weight):
    """The bucket view of ``weight`` once its bucket has looked the same on two consecutive steps, else None."""
    v = _ddp["views"].get(weight.data_ptr())
    if v is None or v.shape != weight.shape or v.stride() != weight.stride() or v.dtype != weight.dtype:
        if _DEBUG:
            with open(_DEBUG + ".rank%d" % torch.distributed.get_rank(), "a") as f:
                f.write("no view for %s strides %s: %s\n" % (tuple(weight.shape), tuple(weight.stride()),
                                                             "unknown parameter" if v is None else
                                                             "view %s strides %s %s" % (tuple(v.shape), tuple(v.stride()), v.dtype)))
        return None
    for (ptr, nbytes), seen in _ddp["buffers"].values():
        if ptr <= v.data_ptr() < ptr + nbytes:
            return v if seen >= 2 else None
    return None


def attach_ddp(ddp, process_group=None):
    """Keep the weight-gradient stream under DistributedDataParallel (see the module text): registers the communication hook."""
    from torch.distributed.algorithms.ddp_comm_hooks import default_hooks
    _ddp["on"] = True
    _ddp["views"].clear()
    _ddp["buffers"].clear()
    STATS["ddp_direct"] = STATS["ddp_inline"] = 0

    def hook(state, bucket):
        bucket_ready(bucket)
        return default_hooks.allreduce_hook(state, bucket)
    ddp.register_comm_hook(process_group, hook)


def detach_ddp():
    _ddp["on"] = False
    _ddp["views"].clear()
    _ddp["buffers"].clear()


# ---- the op-side entry point ---------------------------------------------------------------------------------------------------

def _hooks_are_stream_safe(weight):
    """A tensor hook would see the gradient on the caller's stream; train.Trainer's stride re-labelling hook only makes a view."""
    return not weight._backward_hooks or getattr(weight, "_ud_hooks_stream_safe", False)


def defer(weight, thunk, *keep):
    """dW of ``weight`` = thunk(), computed on the weight-gradient stream when that is safe (see the module text)."""
    if not (ENABLED and weight.is_cuda and weight.is_leaf and weight.grad is None and not torch.is_grad_enabled()
            and _hooks_are_stream_safe(weight) and not getattr(weight, "_post_accumulate_grad_hooks", None)
            and (CAPTURE_OK or not torch.cuda.is_current_stream_capturing())):
        return thunk()      # (a hook on the parameter would read the gradient on the caller's stream)
    view = None
    if _ddp["on"]:
        view = _ddp_view(weight)
        if view is None:
            STATS["ddp_inline"] += 1
            return thunk()
    elif torch.distributed.is_available() and torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1:
        return thunk()      # a process group with peers and no attached reducer: assume a gradient hook this module cannot see
    idx = weight.device.index
    side = _streams.get(idx)
    if side is None:
        side = _streams[idx] = torch.cuda.Stream(weight.device)
    cur = torch.cuda.current_stream(idx)
    task = _current_task()
    if _pending and task != _task[0]:
        join()              # left behind by a backward pass that never reached its callback
    key = weight.data_ptr()
    if key in _deferred:
        # second use inside one backward: autograd adds the two gradients on THIS stream before AccumulateGrad runs
        cur.wait_stream(side)
        STATS["inline_repeat"] += 1
        return thunk()
    side.wait_stream(cur)
    with torch.cuda.stream(side), _lib.workspace_scope(SCOPE[0]):
        g = thunk()
        if view is not None:
            out = view.detach()                         # a fresh alias of the DDP bucket view: nobody else holds THIS tensor
            out.copy_(g)
            g = out
        elif g.dtype != weight.dtype or g.shape != weight.shape or g.stride() != weight.stride():
            out = torch.empty_like(weight)              # the parameter's own strides (preserve_format)
            out.copy_(g)
            g = out
    for t in keep:
        t.record_stream(side)                           # their blocks must not be handed out again before this stream has read them
    if view is None:
        g.record_stream(cur)                            # allocated in the side stream's pool, read by the optimizer on the caller's
    STATS["ddp_direct" if view is not None else "deferred"] += 1
    _deferred.add(key)
    first = not _pending
    _pending.add(idx)
    if first:
        try:
            Variable._execution_engine.queue_callback(join)
            _task[0] = task
        except RuntimeError:                            # not inside an autograd backward pass (a Function's backward called by hand)
            join()
    return g
