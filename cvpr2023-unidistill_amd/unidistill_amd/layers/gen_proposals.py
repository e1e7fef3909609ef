"""Proposal layers of the CenterPoint heads: heat-map top-K decode, range / score filter, rotated NMS.

Mirrors of CenterPointGenProposals and IouAwareGenProposals (reference
unidistill/layers/head/det3d/generate_proposals/centerpoint_gen_proposals.py:8-340,
iou_aware_gen_proposals.py:6-247): same constructor arguments, same ``generate_predicted_boxes``
contract and outputs.  The decode is batched tensor code; NMS runs on ops/nms.py (or any
``nms_fn(boxes, scores, thresh, pre_maxsize, post_max_size)`` passed in -- the CPU tests use the oracle).
"""
import torch
from torch import nn


def _gather_map(feat, inds):
    """feat [B, C, H, W], inds [B, K] (flat y*W+x) -> [B, K, C]."""
    B, C = feat.shape[:2]
    flat = feat.reshape(B, C, -1)
    return flat.gather(2, inds[:, None, :].expand(B, C, inds.shape[1])).transpose(1, 2)


class CenterPointGenProposals(nn.Module):
    def __init__(self, dataset_name, class_names, post_center_limit_range, score_threshold, pc_range,
                 out_size_factor, voxel_size, no_log, nms_iou_threshold_train, nms_pre_max_size_train,
                 nms_post_max_size_train, nms_iou_threshold_test, nms_pre_max_size_test,
                 nms_post_max_size_test, nms_fn=None):
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
        self.nms_fn = nms_fn
        self.training = True

    # -- pieces ------------------------------------------------------------------------------------
    @staticmethod
    def _topk(scores, K):
        """Per-class top-K then top-K over classes (centerpoint_gen_proposals.py:66-83)."""
        B, C, H, W = scores.shape
        s1, i1 = torch.topk(scores.reshape(B, C, -1), K)
        i1 = i1 % (H * W)
        s2, i2 = torch.topk(s1.reshape(B, -1), K)
        cls = (i2 / K).int()
        inds = i1.reshape(B, -1).gather(1, i2)
        ys = (inds / W).int().float()
        xs = (inds % W).int().float()
        return s2, inds, cls, ys, xs

    def _nms_scores(self, scores, inds, task_id, extra):
        return scores

    def _select(self, boxes, scores, thresh, pre, post):
        if self.nms_fn is not None:
            return self.nms_fn(boxes, scores, thresh, pre, post)
        from ..ops import nms
        return nms.nms_rotated(boxes, scores, thresh, pre, post)

    @torch.no_grad()
    def proposal_layer(self, heat, rots, rotc, hei, dim, vel, reg=None, raw_rot=False, task_id=-1, **extra):
        assert reg is not None and raw_rot is False
        B = heat.shape[0]
        K = self.nms_pre_max_size_use
        scores, inds, clses, ys, xs = self._topk(heat, K)
        nms_scores = self._nms_scores(scores, inds, task_id, extra)
        reg = _gather_map(reg, inds)
        xs = xs[:, :, None] + reg[:, :, 0:1]
        ys = ys[:, :, None] + reg[:, :, 1:2]
        rot = torch.atan2(_gather_map(rots, inds), _gather_map(rotc, inds))
        hei = _gather_map(hei, inds)
        dim = _gather_map(dim, inds)
        xs = xs * self.out_size_factor * self.voxel_size[0] + self.pc_range[0]
        ys = ys * self.out_size_factor * self.voxel_size[1] + self.pc_range[1]
        parts = [xs, ys, hei, dim, rot]
        if self.dataset_name == "nuscenes":
            parts.append(_gather_map(vel, inds))
        boxes = torch.cat(parts, dim=2)
        rng = torch.tensor(self.post_center_limit_range, device=boxes.device, dtype=boxes.dtype)
        mask = (boxes[..., :3] >= rng[:3]).all(2) & (boxes[..., :3] <= rng[3:]).all(2)
        mask &= scores > self.score_threshold
        out = []
        for i in range(B):
            m = mask[i]
            b3, sc, lb, ns = boxes[i, m], scores[i, m], clses[i, m].float(), nms_scores[i, m]
            if ns.shape[0] != 0:
                sel = self._select(b3[:, :7], ns, self.nms_iou_threshold_use, self.nms_pre_max_size_use,
                                   self.nms_post_max_size_use)
            else:
                sel = torch.zeros((0,), dtype=torch.long, device=b3.device)
            out.append({"boxes": b3[sel], "scores": sc[sel], "labels": lb[sel].long()})
        return out

    def _task_inputs(self, pred_dict):
        hm = pred_dict["hm"].float().sigmoid()
        dim = pred_dict["dim"].float()
        if not self.no_log:
            dim = torch.clamp(torch.exp(dim), min=0.001, max=30)
        rot = pred_dict["rot"].float()
        vel = pred_dict["vel"].float() if self.dataset_name == "nuscenes" else None
        return dict(heat=hm, rots=rot[:, 0:1], rotc=rot[:, 1:2], hei=pred_dict["height"].float(), dim=dim,
                    vel=vel, reg=pred_dict["reg"].float())

    @torch.no_grad()
    def generate_predicted_boxes(self, forward_ret_dict, data_dict):
        pred_dicts = forward_ret_dict["multi_head_features"]
        phase = "train" if self.training else "test"
        self.nms_iou_threshold_use = getattr(self, f"nms_iou_threshold_{phase}")
        self.nms_pre_max_size_use = getattr(self, f"nms_pre_max_size_{phase}")
        self.nms_post_max_size_use = getattr(self, f"nms_post_max_size_{phase}")
        per_task = []
        for task_id, pred in enumerate(pred_dicts):
            per_task.append(self.proposal_layer(task_id=task_id, **self._task_inputs(pred),
                                                **self._extra_inputs(pred)))
        B = len(per_task[0])
        num_rois = self.nms_post_max_size_use * len(self.class_names)
        out, rois, roi_scores, roi_labels = [], [], [], []
        for b in range(B):
            boxes, scores, labels, offset = [], [], [], 1          # global labels start at 1
            for task_id, names in enumerate(self.class_names):
                boxes.append(per_task[task_id][b]["boxes"])
                scores.append(per_task[task_id][b]["scores"])
                labels.append(per_task[task_id][b]["labels"] + offset)
                offset += len(names)
            boxes, scores, labels = torch.cat(boxes), torch.cat(scores), torch.cat(labels)
            n = boxes.shape[0]
            roi = boxes.new_zeros(num_rois, boxes.shape[-1])
            roi_score, roi_label = scores.new_zeros(num_rois), labels.new_zeros(num_rois)
            roi[:n], roi_score[:n], roi_label[:n] = boxes, scores, labels
            rois.append(roi); roi_scores.append(roi_score); roi_labels.append(roi_label)
            out.append({"pred_boxes": boxes, "pred_scores": scores, "pred_labels": labels})
        data_dict["pred_dicts"] = out
        data_dict["rois"] = torch.stack(rois)
        data_dict["roi_scores"] = torch.stack(roi_scores)
        data_dict["roi_labels"] = torch.stack(roi_labels)
        data_dict["has_class_labels"] = True
        data_dict.pop("batch_index", None)
        return data_dict

    def _extra_inputs(self, pred_dict):
        return {}


class IouAwareGenProposals(CenterPointGenProposals):
    """NMS ranks by score^(1-a) * iou^a with the predicted IoU map (iou_aware_gen_proposals.py:43-66)."""

    def __init__(self, *args, iou_aware_list=None, **kw):
        if iou_aware_list is None and len(args) == 15:
            *args, iou_aware_list = args
        super().__init__(*args, **kw)
        self.iou_aware_list = iou_aware_list

    def _extra_inputs(self, pred_dict):
        return {"iouhm": pred_dict["iou"].float()}

    def _nms_scores(self, scores, inds, task_id, extra):
        B, K = scores.shape
        iou = torch.clamp(_gather_map(extra["iouhm"], inds).reshape(B, K) / 2 + 0.5, 0, 1)
        a = self.iou_aware_list[task_id]
        return (scores ** (1 - a)).mul(iou ** a)
