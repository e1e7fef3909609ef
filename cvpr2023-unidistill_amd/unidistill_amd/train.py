"""Cross-modality distillation training step (student + frozen teacher), without Lightning.

Mirrors Exp.__init__ / Exp.training_step / configure_optimizers of the four distill experiments
(e.g. BEVFusion_nuscenes_centerhead_camera_exp_distill_lidar.py:388-513 and
BEVFusion_nuscenes_base_exp.py:436-441; trainer flags base_cli.py:40-45): AdamW(lr 2e-4,
wd 1e-7), grad-norm clip 0.1, loss = rpn + w_feat*feat + w_rel*rel + w_resp*(cls + reg).
Host-side work of the reference's step (valid-box python loop with one sync per box, numpy box
corners, numpy gaussian mask, reloading the teacher state_dict every step) is replaced by device
kernels; data parallelism is one process per GPU with DistributedDataParallel over RCCL.
"""
import torch
from torch import nn

from . import config as C
from .dist import get_world_size, reduce_mean_many
from .models import BEVFusionCenterHead
from .ops import distill as D


def build_model(modality, **kw):
    """modality: 'camera' | 'lidar' | 'fusion' -> BEVFusionCenterHead."""
    cfg = C.model_cfg(lidar=modality in ("lidar", "fusion"), camera=modality in ("camera", "fusion"))
    return BEVFusionCenterHead(cfg, **kw)


class DistillStep(nn.Module):
    """Holds the trainable student and the frozen teacher; forward(batch) returns the loss dict."""

    def __init__(self, experiment="camera_exp_distill_lidar", teacher_train_mode=False,
                 student=None, teacher=None):
        super().__init__()
        e = dict(C.DISTILL_EXPERIMENTS[experiment]) if isinstance(experiment, str) else dict(experiment)
        self.exp = e
        self.model = student if student is not None else build_model(e["student"])
        self.teacher_model = teacher if teacher is not None else build_model(e["teacher"])
        self.teacher_model.det_head.dense_head.distill = True
        for p in self.teacher_model.parameters():
            p.requires_grad = False
        # Lightning's model.train() flips the registered teacher back to train
        # mode (BN batch statistics); teacher_train_mode reproduces that, default is eval.
        self.teacher_train_mode = teacher_train_mode

    def train(self, mode=True):
        super().train(mode)
        self.teacher_model.train(mode and self.teacher_train_mode)
        return self

    def forward(self, batch):
        points = batch.get("points")
        if points is not None and not isinstance(points, (list, tuple)):
            points = [p for p in points]
        imgs, metas = batch.get("imgs"), batch.get("mats_dict")
        gt9 = batch["gt_boxes"]
        gt = torch.cat([gt9, (batch["gt_labels"] + 1).unsqueeze(2)], 2)
        e = self.exp
        ret, tb, feat_s, bev_s, resp_s, _ = self.model(points, imgs, metas, gt)
        with torch.no_grad():
            feat_t, bev_t, resp_t = self.teacher_model(points, imgs, metas, gt, return_feature=True)
        corners, valid = D.box_corners_bev(gt9, C.POINT_CLOUD_RANGE, C.VOXEL_SIZE, C.OUT_SIZE_FACTOR)
        mask = D.calculate_box_mask_gaussian(resp_s[0]["reg"].shape, gt, C.POINT_CLOUD_RANGE,
                                             C.VOXEL_SIZE, C.OUT_SIZE_FACTOR)
        w_box, w_mask = reduce_mean_many([valid.float().sum(), mask.sum()])   # one collective
        loss_feat = D.FeatureDistillLoss(feat_s, feat_t, corners, valid, weight=w_box)
        loss_rel = D.BEVDistillLoss(bev_s, bev_t, corners, valid, weight=w_box)
        loss_cls, loss_reg = D.ResponseDistillLoss(resp_s, resp_t, gt, C.POINT_CLOUD_RANGE,
                                                   C.VOXEL_SIZE, C.OUT_SIZE_FACTOR, clamp=e["clamp"],
                                                   weight=w_mask, mask=mask)
        loss = ret["loss"].mean() + e["feat"] * loss_feat + e["rel"] * loss_rel \
            + e["resp"] * (loss_cls + loss_reg)
        tb.update(loss_feature=loss_feat.detach(), loss_bev_rel=loss_rel.detach(),
                  loss_resp_cls=loss_cls.detach(), loss_resp_reg=loss_reg.detach())
        return {"loss": loss, "tb": tb}


class Trainer:
    """AdamW + grad clip + (optional) DDP around a module whose forward(batch) returns {'loss'}."""

    def __init__(self, step_module, lr=2e-4, weight_decay=1e-7, grad_clip=0.1, device=None,
                 bucket_cap_mb=64, autocast_dtype=None):
        self.device = device or torch.device("cuda", torch.cuda.current_device())
        self.module = step_module.to(self.device)
        self.module.train()
        trainable = [p for p in self.module.parameters() if p.requires_grad]
        self.ddp = None
        if get_world_size() > 1:
            # gradients are all-reduced over RCCL/xGMI in ~64 MB buckets, overlapped with backward
            self.ddp = nn.parallel.DistributedDataParallel(
                self.module, device_ids=[self.device.index], bucket_cap_mb=bucket_cap_mb,
                gradient_as_bucket_view=True, broadcast_buffers=False, find_unused_parameters=False)
        self.opt = torch.optim.AdamW(trainable, lr=lr, weight_decay=weight_decay, fused=True)
        self.params = trainable
        self.grad_clip = grad_clip
        self.autocast_dtype = autocast_dtype

    def step(self, batch):
        self.opt.zero_grad(set_to_none=True)
        fn = self.ddp if self.ddp is not None else self.module
        if self.autocast_dtype is not None:
            with torch.autocast("cuda", dtype=self.autocast_dtype):
                out = fn(batch)
        else:
            out = fn(batch)
        out["loss"].backward()
        if self.grad_clip:
            torch.nn.utils.clip_grad_norm_(self.params, self.grad_clip, foreach=True)
        self.opt.step()
        return out


class DetectStep(nn.Module):
    """Plain (non-distill) detector training step: Exp.training_step of the base experiments
    (BEVFusion_nuscenes_base_exp.py:360-375)."""

    def __init__(self, modality="lidar", model=None):
        super().__init__()
        self.model = model if model is not None else build_model(modality)

    def forward(self, batch):
        points = batch.get("points")
        if points is not None and not isinstance(points, (list, tuple)):
            points = [p for p in points]
        gt = torch.cat([batch["gt_boxes"], (batch["gt_labels"] + 1).unsqueeze(2)], 2)
        ret, tb, *_ = self.model(points, batch.get("imgs"), batch.get("mats_dict"), gt)
        return {"loss": ret["loss"].mean(), "tb": tb}


def synthetic_batch(device, batch_size=1, rank=0, ncam=6, sweeps=1, n_boxes=40, max_boxes=50,
                    with_points=True, with_imgs=True, seed=1234):
    """collate_fn-shaped synthetic batch on the device."""
    import numpy as np
    from . import synthetic as syn
    g = syn.rng(seed, rank)
    batch = {}
    if with_points:
        clouds = [syn.lidar_cloud(g, 30000, sweeps) for _ in range(batch_size)]
        batch["points"] = torch.from_numpy(syn.pad_clouds(clouds)).to(device)
    if with_imgs:
        H, W = C.IMG_DIM
        batch["imgs"] = torch.from_numpy(
            g.standard_normal((batch_size, 1, ncam, 3, H, W)).astype(np.float32)).to(device)
        s2e, intr, ida, bda = syn.camera_rig(g, batch_size, ncam)
        batch["mats_dict"] = {"sensor2ego_mats": torch.from_numpy(s2e).to(device),
                              "intrin_mats": torch.from_numpy(intr).to(device),
                              "ida_mats": torch.from_numpy(ida).to(device),
                              "sensor2sensor_mats": torch.from_numpy(np.tile(np.eye(4, dtype=np.float32),
                                                                             (batch_size, 1, ncam, 1, 1))).to(device),
                              "bda_mat": torch.from_numpy(bda).to(device)}
    boxes, labels = syn.gt_boxes(g, batch_size, n_boxes, max_boxes)
    batch["gt_boxes"] = torch.from_numpy(boxes).to(device)
    batch["gt_labels"] = torch.from_numpy(labels).to(device)
    return batch
