"""Proposal layers of the CenterPoint heads: heat-map top-K decode, range / score filter, rotated NMS.

Mirrors of CenterPointGenProposals and IouAwareGenProposals (reference
unidistill/layers/head/det3d/generate_proposals/centerpoint_gen_proposals.py:8-340,
iou_aware_gen_proposals.py:6-247): same constructor arguments, same ``generate_predicted_boxes``
contract and outputs.  The whole layer -- every task and every sample -- is three launches of
libunidistill_hip (``ud_proposal_layer``, csrc/proposals.hip): the head tensors are read in place
through their strides (NCHW planes or channel slices of the packed channels-last head output), and
the only host read is the per-sample box count that sizes the ``pred_dicts`` slices.  GPU only.
"""
import ctypes

import torch
from torch import nn

from .. import _lib

_HEADS = ("hm", "reg", "height", "dim", "rot", "vel", "iou")


def _desc(t, H, W):
    """(device pointer, (batch, channel, pixel) strides in elements) of a logical [B, c, H, W] fp32 tensor
    whose H, W dims collapse to one pixel stride (true for NCHW and channels-last alike)."""
    sb, sc, sh, sw = t.stride()
    if t.dtype != torch.float32 or (H > 1 and sh != W * sw):
        t = t.float().contiguous()
        sb, sc, sh, sw = t.stride()
    return t, (sb, sc, sw)


class CenterPointGenProposals(nn.Module):
    iou_aware = False

    def __init__(self, dataset_name, class_names, post_center_limit_range, score_threshold, pc_range,
                 out_size_factor, voxel_size, no_log, nms_iou_threshold_train, nms_pre_max_size_train,
                 nms_post_max_size_train, nms_iou_threshold_test, nms_pre_max_size_test,
                 nms_post_max_size_test):
        super().__init__()
        self.dataset_name = dataset_name
        self.class_names = class_names
        self.post_center_limit_range = post_center_limit_range
        self.score_threshold = score_threshold
        self.pc_range = pc_range
        self.out_size_factor = out_size_factor
        self.voxel_size = voxel_size
        self.no_log = no_log
        self.nms_iou_threshold_train = nms_iou_threshold_train
        self.nms_pre_max_size_train = nms_pre_max_size_train
        self.nms_post_max_size_train = nms_post_max_size_train
        self.nms_iou_threshold_test = nms_iou_threshold_test
        self.nms_pre_max_size_test = nms_pre_max_size_test
        self.nms_post_max_size_test = nms_post_max_size_test
        self.training = True

    def _alphas(self, n_tasks):
        return None

    def _run(self, pred_dicts, class_offsets, alphas):
        """All tasks of ``pred_dicts`` in one ud_proposal_layer call -> (rois, scores, labels, counts)."""
        lib = _lib.load()
        T = len(pred_dicts)
        hm0 = pred_dicts[0]["hm"]
        _lib.require_gpu(hm0)
        B, _, H, W = hm0.shape
        nbox = 9 if self.dataset_name == "nuscenes" else 7
        ptrs, strides, keep_alive, ncs = [], [], [], []
        for pred in pred_dicts:
            ncs.append(int(pred["hm"].shape[1]))
            for name in _HEADS:
                t = pred.get(name)
                if t is None or (name == "vel" and nbox == 7) or (name == "iou" and not self.iou_aware):
                    ptrs.append(None)
                    strides += [0, 0, 0]
                    continue
                _lib.require_gpu(t)
                t, st = _desc(t, H, W)
                keep_alive.append(t)
                ptrs.append(t.data_ptr())
                strides += list(st)
        K, post = int(self.nms_pre_max_size_use), int(self.nms_post_max_size_use)
        dev = hm0.device
        rois = torch.empty((B, T * post, nbox), dtype=torch.float32, device=dev)
        scores = torch.empty((B, T * post), dtype=torch.float32, device=dev)
        labels = torch.empty((B, T * post), dtype=torch.int64, device=dev)
        counts = torch.empty((B,), dtype=torch.int32, device=dev)
        nbytes = lib.ud_proposal_workspace_bytes(B, T, K)
        if nbytes == 0:
            raise RuntimeError(f"ud_proposal_layer: unsupported sizes B={B} T={T} K={K}")
        ws = _lib.workspace(dev, nbytes, "proposals")
        c_ptrs = (ctypes.c_void_p * len(ptrs))(*ptrs)
        c_str = (ctypes.c_longlong * len(strides))(*strides)
        c_nc = (ctypes.c_int * T)(*ncs)
        c_off = (ctypes.c_int * T)(*class_offsets)
        c_alpha = (ctypes.c_float * T)(*alphas) if alphas is not None else None
        c_rng = (ctypes.c_float * 6)(*[float(v) for v in self.post_center_limit_range])
        _lib.check(lib.ud_proposal_layer(
            c_ptrs, c_str, c_nc, c_off, c_alpha, B, T, H, W, K, post, nbox, 1 if self.no_log else 0,
            float(self.out_size_factor), float(self.voxel_size[0]), float(self.voxel_size[1]),
            float(self.pc_range[0]), float(self.pc_range[1]), c_rng, float(self.score_threshold),
            float(self.nms_iou_threshold_use), _lib.ptr(rois), _lib.ptr(scores), _lib.ptr(labels),
            _lib.ptr(counts), _lib.ptr(ws), ws.numel(), _lib.stream_of(hm0)), "ud_proposal_layer")
        return rois, scores, labels, counts

    def _phase(self):
        phase = "train" if self.training else "test"
        self.nms_iou_threshold_use = getattr(self, f"nms_iou_threshold_{phase}")
        self.nms_pre_max_size_use = getattr(self, f"nms_pre_max_size_{phase}")
        self.nms_post_max_size_use = getattr(self, f"nms_post_max_size_{phase}")

    @torch.no_grad()
    def proposal_layer(self, pred_dict, task_id=-1):
        """One task (reference proposal_layer, iou_aware_gen_proposals.py:43-139): list over samples of
        {"boxes", "scores", "labels"} with task-local labels."""
        if not hasattr(self, "nms_pre_max_size_use"):
            self._phase()
        alphas = self._alphas(max(task_id + 1, 1))
        rois, scores, labels, counts = self._run([pred_dict], [-1], alphas and [alphas[task_id]])
        counts = counts.tolist()
        return [{"boxes": rois[b, :n], "scores": scores[b, :n], "labels": labels[b, :n]}
                for b, n in enumerate(counts)]

    @torch.no_grad()
    def generate_predicted_boxes(self, forward_ret_dict, data_dict):
        pred_dicts = forward_ret_dict["multi_head_features"]
        self._phase()
        offsets, off = [], 0                                   # global labels start at 1 (added on the device)
        for names in self.class_names:
            offsets.append(off)
            off += len(names)
        rois, scores, labels, counts = self._run(pred_dicts, offsets, self._alphas(len(pred_dicts)))
        counts = counts.tolist()                               # the one host read of the layer
        # independent tensors, as the reference returns them (in-place post-processing of pred_boxes must not reach rois)
        data_dict["pred_dicts"] = [{"pred_boxes": rois[b, :n].clone(), "pred_scores": scores[b, :n].clone(),
                                    "pred_labels": labels[b, :n].clone()} for b, n in enumerate(counts)]
        data_dict["rois"] = rois
        data_dict["roi_scores"] = scores
        data_dict["roi_labels"] = labels
        data_dict["has_class_labels"] = True
        data_dict.pop("batch_index", None)
        return data_dict


class IouAwareGenProposals(CenterPointGenProposals):
    """NMS ranks by score^(1-a) * iou^a with the predicted IoU map (iou_aware_gen_proposals.py:43-66)."""
    iou_aware = True

    def __init__(self, *args, iou_aware_list=None, **kw):
        if iou_aware_list is None and len(args) == 15:
            *args, iou_aware_list = args
        super().__init__(*args, **kw)
        self.iou_aware_list = iou_aware_list

    def _alphas(self, n_tasks):
        return [float(self.iou_aware_list[t]) for t in range(n_tasks)]
