"""Rotated-BEV IoU and NMS on libunidistill_hip (ud_nms_rotated_bev, ud_boxes_iou_bev).

``nms_gpu`` / ``boxes_iou_bev_gpu`` keep the call signatures of the reference's missing
``iou3d_nms_cuda`` extension (unidistill/layers/head/det3d/generate_proposals/
centerpoint_gen_proposals.py:85-105); ``nms_rotated`` is the device-resident form the proposal
layer here uses (no host round trip for the keep list).
"""
import torch

from .. import _lib


def _run(boxes, thresh):
    """boxes f32[N,7] on the GPU, sorted by descending score -> (keep i64[N] padded with -1, count i32[1])."""
    lib = _lib.load()
    _lib.require_gpu(boxes)
    boxes = boxes.contiguous().float()
    n = boxes.shape[0]
    keep = torch.empty((max(n, 1),), dtype=torch.int64, device=boxes.device)
    count = torch.zeros((1,), dtype=torch.int32, device=boxes.device)
    ws = _lib.workspace(boxes.device, lib.ud_nms_bev_workspace_bytes(max(n, 1)), "nms")
    _lib.check(lib.ud_nms_rotated_bev(_lib.ptr(boxes), n, float(thresh), _lib.ptr(keep), _lib.ptr(count),
                                      _lib.ptr(ws), ws.numel(), _lib.stream_of(boxes)), "ud_nms_rotated_bev")
    return keep[:n], count


def nms_gpu(boxes, keep, thresh):
    """iou3d_nms_cuda.nms_gpu(boxes, keep, thresh): fills the (CPU) LongTensor ``keep`` with the kept
    indices of the score-sorted ``boxes`` [N, 7] and returns their number."""
    kept, count = _run(boxes[:, :7], thresh)
    n = int(count.item())
    keep[:n] = kept[:n].to(keep.device)
    return n


def nms_rotated(boxes, scores, thresh, pre_maxsize=None, post_max_size=None):
    """_nms_gpu_3d of the reference (centerpoint_gen_proposals.py:85-105): indices into ``boxes`` of the
    kept boxes, by descending score."""
    order = scores.sort(0, descending=True)[1]
    if pre_maxsize is not None:
        order = order[:pre_maxsize]
    kept, count = _run(boxes[order][:, :7], thresh)
    n = int(count.item())
    selected = order[kept[:n]].contiguous()
    if post_max_size is not None:
        selected = selected[:post_max_size]
    return selected


def boxes_iou_bev_gpu(boxes_a, boxes_b):
    """[Na, Nb] IoU of the BEV footprints of boxes (x, y, z, dx, dy, dz, heading)."""
    _lib.require_gpu(boxes_a, boxes_b)
    a, b = boxes_a[:, :7].contiguous().float(), boxes_b[:, :7].contiguous().float()
    out = torch.empty((a.shape[0], b.shape[0]), dtype=torch.float32, device=a.device)
    _lib.check(_lib.load().ud_boxes_iou_bev(_lib.ptr(a), a.shape[0], _lib.ptr(b), b.shape[0], _lib.ptr(out),
                                            _lib.stream_of(a)), "ud_boxes_iou_bev")
    return out
