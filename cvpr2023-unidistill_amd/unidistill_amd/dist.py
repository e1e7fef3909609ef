"""Collective helpers: mirror of unidistill/utils/torch_dist.py:1-64 on torch.distributed
(backend "nccl" == RCCL over xGMI on ROCm), plus a batched variant.

The reference calls reduce_mean on 27 separate 1-element tensors per training step, each followed
by a host sync (losses/det3d.py:313,353,414; center_head_iou_aware.py:285; distill files
:243,283,382).  ``reduce_mean_many`` packs any number of scalars into ONE all-reduce.
"""
import torch
from torch import distributed as dist


def is_available():
    return dist.is_available()


def is_distributed():
    return dist.is_available() and dist.is_initialized()


def get_rank():
    return dist.get_rank() if is_distributed() else 0


def get_world_size():
    return dist.get_world_size() if is_distributed() else 1


def synchronize():
    if is_distributed() and dist.get_world_size() > 1:
        dist.barrier()


def reduce_sum(tensor):
    if get_world_size() < 2:
        return tensor
    tensor = tensor.clone()
    dist.all_reduce(tensor, op=dist.ReduceOp.SUM)
    return tensor


def reduce_mean(tensor):
    return reduce_sum(tensor) / float(get_world_size())


def reduce_mean_many(scalars):
    """[0-dim tensors] -> list of their cross-rank means, with a single collective."""
    if get_world_size() < 2:
        return list(scalars)
    packed = torch.stack([s.detach().reshape(()).float() for s in scalars])
    dist.all_reduce(packed, op=dist.ReduceOp.SUM)
    packed = packed / float(get_world_size())
    return list(packed.unbind(0))


def all_gather_object(obj):
    if get_world_size() < 2:
        return [obj]
    out = [None for _ in range(get_world_size())]
    dist.all_gather_object(out, obj)
    return out
