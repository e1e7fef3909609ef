"""Cross-modality distillation training step (student + frozen teacher), without Lightning.

Mirrors Exp.__init__ / Exp.training_step / configure_optimizers of the four distill experiments
(e.g. BEVFusion_nuscenes_centerhead_camera_exp_distill_lidar.py:388-513 and
BEVFusion_nuscenes_base_exp.py:436-441; trainer flags base_cli.py:40-45): AdamW(lr 2e-4,
wd 1e-7), grad-norm clip 0.1, loss = rpn + w_feat*feat + w_rel*rel + w_resp*(cls + reg).
Host-side work of the reference's step (valid-box python loop with one sync per box, numpy box
corners, numpy gaussian mask, reloading the teacher state_dict every step) is replaced by device
kernels; data parallelism is one process per GPU with DistributedDataParallel over RCCL.
"""
import os

import torch
from torch import nn

from . import config as C
from . import _lib
from .dist import get_world_size, is_distributed, reduce_mean_many
from .models import BEVFusionCenterHead
from .ops import distill as D, wgrad_stream


def build_model(modality, **kw):
    """modality: 'camera' | 'lidar' | 'fusion' -> BEVFusionCenterHead."""
    cfg = C.model_cfg(lidar=modality in ("lidar", "fusion"), camera=modality in ("camera", "fusion"))
    return BEVFusionCenterHead(cfg, **kw)


def _tensors_in(obj):
    """All tensors inside nested lists / tuples / dicts."""
    if torch.is_tensor(obj):
        yield obj
    elif isinstance(obj, dict):
        for v in obj.values():
            yield from _tensors_in(v)
    elif isinstance(obj, (list, tuple)):
        for v in obj:
            yield from _tensors_in(v)


class DistillStep(nn.Module):
    """Holds the trainable student and the frozen teacher; forward(batch) returns the loss dict."""

    def __init__(self, experiment="camera_exp_distill_lidar", teacher_train_mode=False,
                 student=None, teacher=None, geometry=None):
        super().__init__()
        e = dict(C.DISTILL_EXPERIMENTS[experiment]) if isinstance(experiment, str) else dict(experiment)
        self.exp = e
        # the experiment modules' _POINT_CLOUD_RANGE / _VOXEL_SIZE / _OUT_SIZE_FACTOR / _GRID_SIZE constants
        # (distill_lidar.py:33-50); overridable so shrunk configurations can be pinned against the reference
        geo = dict(point_cloud_range=C.POINT_CLOUD_RANGE, voxel_size=C.VOXEL_SIZE,
                   out_size_factor=C.OUT_SIZE_FACTOR, grid_size=C.GRID_SIZE)
        geo.update(geometry or {})
        self.geo = geo
        self.model = student if student is not None else build_model(e["student"])
        self.teacher_model = teacher if teacher is not None else build_model(e["teacher"])
        self.teacher_model.det_head.dense_head.distill = True
        for p in self.teacher_model.parameters():
            p.requires_grad = False
        # Lightning's model.train() flips the registered teacher back to train
        # mode (BN batch statistics); teacher_train_mode reproduces that, default is eval.
        self.teacher_train_mode = teacher_train_mode
        self.teacher_model.train(self.training and teacher_train_mode)   # frozen teacher: eval from the start
        # the reference loads the teacher with self.teacher_model.load_state_dict(ckpt) (..._distill_lidar.py:424): a (late) load
        # into the submodule must not be undone by a snapshot taken before it
        self.teacher_model.register_load_state_dict_post_hook(self._drop_teacher_snapshot)

    def train(self, mode=True):
        super().train(mode)
        self.teacher_model.train(mode and self.teacher_train_mode)
        return self

    # The step is split into phases so that neither the network pass nor its backward contains a
    # collective or a host sync:
    #   prep (gt-only work + local normalisers) -> reduce (ONE all-reduce) -> teacher -> student.
    @staticmethod
    def _points(batch):
        points = batch.get("points")
        if points is not None and not isinstance(points, (list, tuple)):
            points = [p for p in points]
        return points

    def prep(self, batch):
        """Everything that depends on the GT boxes only: FCOS targets, BEV box corners / validity,
        gaussian mask, and the 14 loss normalisers of this rank packed into one tensor."""
        gt9 = batch["gt_boxes"]
        gt = torch.cat([gt9, (batch["gt_labels"] + 1).unsqueeze(2)], 2)
        head = self.model.det_head.dense_head
        targets = head.assign_targets(gt)
        for enc in targets["box_encoding"].values():
            enc[torch.isinf(enc)] = 0
        targets["box_encoding_clean"] = True       # DetHead.forward skips its own clean-up only on this flag
        G = self.geo
        pcr, vs, osf = G["point_cloud_range"], G["voxel_size"], G["out_size_factor"]
        corners, valid = D.box_corners_bev(gt9, pcr, vs, osf)
        mask = D.calculate_box_mask_gaussian((gt.shape[0], 1, G["grid_size"][1] // osf, G["grid_size"][0] // osf),
                                             gt, pcr, vs, osf)
        local = torch.stack(head.local_normalisers(targets) + [valid.float().sum(), mask.sum()])
        return {"gt": gt, "targets": targets, "corners": corners, "valid": valid, "mask": mask,
                "local": local}

    @staticmethod
    def reduce(prep):
        """Global mean of the packed normalisers: the step's only loss-side collective."""
        norm = prep["local"]
        if get_world_size() > 1:
            norm = norm.clone()
            torch.distributed.all_reduce(norm, op=torch.distributed.ReduceOp.SUM)
            norm = norm / float(get_world_size())
        prep["norm"] = norm
        return prep

    def _reset_teacher_buffers(self):
        """teacher_train_mode: the reference reloads the teacher's checkpoint before every teacher pass
        (training_step, ..._distill_lidar.py:463: load_state_dict(self.checkpoint_state_dict)), which undoes the running-statistics
        update of the previous step's train-mode BatchNorms.  Here: a snapshot of the buffers copied back with one foreach launch.
        The snapshot is the state the buffers had when they were last set from OUTSIDE the step: it is retaken when anybody but
        the teacher pass itself has written them since (a checkpoint loaded into self.teacher_model or into this module, .to(), a
        manual edit) -- detected by the buffers' version counters, which the statistics kernels bump (ops/bn_act.py) -- and
        dropped by the load_state_dict hooks below."""
        bufs = [b for b in self.teacher_model.buffers()]
        snap = getattr(self, "_teacher_snapshot", None)
        seen = getattr(self, "_teacher_versions", None)
        stale = (snap is None or len(snap) != len(bufs) or seen is None
                 or any(a.device != b.device or a.shape != b.shape or a.dtype != b.dtype for a, b in zip(snap, bufs))
                 or any(b._version != v for b, v in zip(bufs, seen)))
        if stale:
            self._teacher_snapshot = [b.detach().clone() for b in bufs]
        else:
            torch._foreach_copy_(bufs, snap)
        return bufs

    def _note_teacher_versions(self, bufs):
        self._teacher_versions = [b._version for b in bufs]

    def _drop_teacher_snapshot(self, *_):
        self._teacher_snapshot = None
        self._teacher_versions = None

    def load_state_dict(self, *a, **k):
        self._drop_teacher_snapshot()
        return super().load_state_dict(*a, **k)

    @torch.no_grad()
    def teacher(self, batch, prep, lidar_prepared=None):
        bufs = self._reset_teacher_buffers() if (self.teacher_train_mode and self.teacher_model.training) else None
        out = self.teacher_model(self._points(batch), batch.get("imgs"), batch.get("mats_dict"),
                                 prep["gt"], return_feature=True, lidar_prepared=lidar_prepared)
        if bufs is not None:
            self._note_teacher_versions(bufs)      # what the teacher pass itself left: anything newer came from outside
        return out

    @torch.no_grad()
    def teacher_geometry(self, batch):
        """The part of a LiDAR / fusion teacher's pass that reads sizes back to the host (voxel and site counts)."""
        enc = self.teacher_model.lidar_encoder
        return None if enc is None else enc.prepare(self._points(batch))

    def student_loss(self, batch, prep, teacher_out, join=None):
        e = self.exp
        norm = prep["norm"]
        nh = norm.numel() - 2
        ret, tb, feat_s, bev_s, resp_s, _ = self.model(
            self._points(batch), batch.get("imgs"), batch.get("mats_dict"), prep["gt"],
            targets=prep["targets"], loss_norm=list(norm[:nh].unbind(0)))
        if callable(teacher_out):                  # teacher enqueued AFTER the student forward (see forward())
            teacher_out, join = teacher_out()
        if join is not None:                       # teacher stream -> main stream hand-over
            torch.cuda.current_stream(prep["gt"].device).wait_stream(join)
        feat_t, bev_t, resp_t = teacher_out
        w_box, w_mask = norm[nh], norm[nh + 1]
        loss_feat = D.FeatureDistillLoss(feat_s, feat_t, prep["corners"], prep["valid"], weight=w_box)
        loss_rel = D.BEVDistillLoss(bev_s, bev_t, prep["corners"], prep["valid"], weight=w_box)
        loss_cls, loss_reg = D.ResponseDistillLoss(resp_s, resp_t, prep["gt"], self.geo["point_cloud_range"],
                                                   self.geo["voxel_size"], self.geo["out_size_factor"],
                                                   clamp=e["clamp"], weight=w_mask, mask=prep["mask"])
        loss = ret["loss"].mean() + e["feat"] * loss_feat + e["rel"] * loss_rel \
            + e["resp"] * (loss_cls + loss_reg)
        tb.update(loss_feature=loss_feat.detach(), loss_bev_rel=loss_rel.detach(),
                  loss_resp_cls=loss_cls.detach(), loss_resp_reg=loss_reg.detach())
        return {"loss": loss, "tb": tb}

    overlap_teacher = True      # frozen teacher on a second HIP stream, concurrent with the student forward
    # teacher's size reads before the student forward is enqueued: measured SLOWER (fp32 88.2 -> 90.1, bf16 27.1 -> 30.1 ms
    # per step): the host then blocks at the top of the step, while the GPU still works on the previous one, instead of
    # behind 30+ ms of freshly queued student work.  Kept as an A/B switch.
    hoist_teacher_geometry = os.environ.get("UD_HOIST_GEOMETRY", "0") == "1"
    teacher_first = os.environ.get("UD_TEACHER_FIRST", "0") == "1"   # enqueue order of the two forwards (see forward())

    def forward(self, batch):
        prep = self.reduce(self.prep(batch))
        gt = prep["gt"]
        if not (self.overlap_teacher and gt.is_cuda):
            return self.student_loss(batch, prep, self.teacher(batch, prep))
        # The teacher and the student forward are independent until the distillation losses, and both
        # are chains of short dependent kernels: on two streams the GPU fills one chain's gaps with the
        # other's kernels.  The teacher gets its own scratch buffers (ops on different streams must not
        # share a workspace) and its outputs are handed to the main stream explicitly.
        cur = torch.cuda.current_stream(gt.device)
        side = getattr(self, "_teacher_stream", None)
        if side is None or side.device != gt.device:
            side = self._teacher_stream = torch.cuda.Stream(gt.device)
        ready = cur.record_event()                 # the batch and the GT-side tensors are ready from here on
        side.wait_event(ready)
        with torch.cuda.stream(side), _lib.workspace_scope("teacher_stream"):
            lidar_prepared = self.teacher_geometry(batch) if self.hoist_teacher_geometry else None

        def run_teacher():
            with torch.cuda.stream(side), _lib.workspace_scope("teacher_stream"):
                tout = self.teacher(batch, prep, lidar_prepared)
            for part in tout:
                for t in _tensors_in(part):
                    t.record_stream(cur)
            return tout, side
        if self.teacher_first:
            tout, _ = run_teacher()
            return self.student_loss(batch, prep, tout, join=side)
        # Student forward first: a LiDAR / fusion teacher reads five sizes back to the host (voxel and site counts), and
        # every read waits for the side stream.  With the student's forward already queued on the main stream the GPU
        # stays busy during those waits; with the teacher first it ran the teacher's small kernels alone and then idled
        # at the distillation losses while the host caught up (3.3 ms of idle per fp32 step in the trace).
        return self.student_loss(batch, prep, run_teacher)


def to_channels_last(module):
    """NHWC weights for every 2-D conv / deconv / BN (MIOpen's implicit-GEMM kernels are NHWC
    native; NCHW tensors get transposed around every call).  Sparse-conv weights (rank 5) and
    everything else are left alone."""
    for m in module.modules():
        if isinstance(m, (nn.Conv2d, nn.ConvTranspose2d)) and m.weight.shape[2] * m.weight.shape[3] > 1:
            # (1x1 kernels are layout-ambiguous; converting them only confuses DDP's bucket views)
            m.weight.data = m.weight.data.contiguous(memory_format=torch.channels_last)
    return module


class Trainer:
    """AdamW + MultiStepLR + grad clip + (optional) DDP around a module whose forward(batch) returns {'loss'}.

    configure_optimizers of the reference (BEVFusion_nuscenes_base_exp.py:436-441) returns
    ``[AdamW(lr, weight_decay=1e-7)], [MultiStepLR(optimizer, [10, 15])]``; Lightning steps such a scheduler
    once per EPOCH: ``epoch_end()`` here is that hook (``lr_milestones=None`` disables it)."""

    def __init__(self, step_module, lr=2e-4, weight_decay=1e-7, grad_clip=0.1, device=None,
                 bucket_cap_mb=64, autocast_dtype=None, channels_last=False, lr_milestones=(10, 15),
                 lr_gamma=0.1):
        self.device = device or torch.device("cuda", torch.cuda.current_device())
        self.module = step_module.to(self.device)
        if channels_last:
            to_channels_last(self.module)
        self.module.train()
        trainable = [p for p in self.module.parameters() if p.requires_grad]
        self.ddp = None
        # UD_FORCE_DDP=1: wrap even a one-rank process group (the RCCL path exercised on a one-GPU box)
        if get_world_size() > 1 or (is_distributed() and os.environ.get("UD_FORCE_DDP") == "1"):
            # DDP compares gradient strides with its bucket views literally; a 1x1 (transposed) convolution's weight
            # gradient comes back from the channels-last kernels with different strides on its size-1 axes (same
            # memory), which would cost a warning and a copy per step: re-label the strides, no data movement.
            for p in trainable:
                if p.dim() == 4 and p.shape[2] == 1 and p.shape[3] == 1 and p.is_contiguous():
                    p.register_hook(lambda g, st=p.stride(): g.as_strided(g.shape, st)
                                    if (g.stride() != st and g.is_contiguous()) else g)
                    p._ud_hooks_stream_safe = True      # a view, no kernel: ops/wgrad_stream.py may still defer this gradient
            # gradients are all-reduced over RCCL/xGMI in ~64 MB buckets, overlapped with backward
            self.ddp = nn.parallel.DistributedDataParallel(
                self.module, device_ids=[self.device.index] if self.device.type == "cuda" else None,
                output_device=self.device.index if self.device.type == "cuda" else None,
                bucket_cap_mb=bucket_cap_mb,
                gradient_as_bucket_view=True, broadcast_buffers=False, find_unused_parameters=False)
            # DDP's reducer reads each gradient on the autograd stream right after AccumulateGrad: the weight-gradient stream
            # stays on by writing into the bucket views and joining per bucket in a communication hook (ops/wgrad_stream.py)
            if wgrad_stream.ENABLED and self.device.type == "cuda":
                wgrad_stream.attach_ddp(self.ddp)
        self.opt = torch.optim.AdamW(trainable, lr=lr, weight_decay=weight_decay, fused=True)
        self.scheduler = None
        if lr_milestones:
            self.scheduler = torch.optim.lr_scheduler.MultiStepLR(self.opt, list(lr_milestones), gamma=lr_gamma)
        self.epoch = 0
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
        wgrad_stream.join()                        # (already done by the engine callback of ops/wgrad_stream.py; idempotent)
        if self.grad_clip:
            torch.nn.utils.clip_grad_norm_(self.params, self.grad_clip, foreach=True)
        self.opt.step()
        return out

    def epoch_end(self):
        """Lightning's per-epoch scheduler step: lr x 0.1 after epochs 10 and 15 (MultiStepLR [10, 15])."""
        self.epoch += 1
        if self.scheduler is not None:
            self.scheduler.step()
        return self.opt.param_groups[0]["lr"]

    def state_dict(self):
        return {"optimizer": self.opt.state_dict(), "epoch": self.epoch,
                "scheduler": None if self.scheduler is None else self.scheduler.state_dict()}

    def load_state_dict(self, state):
        from .ops import invalidate_caches
        invalidate_caches(self.module)          # cached re-layouts of frozen weights never outlive a (re)load
        self.opt.load_state_dict(state["optimizer"])
        self.epoch = state.get("epoch", 0)
        if self.scheduler is not None and state.get("scheduler") is not None:
            self.scheduler.load_state_dict(state["scheduler"])


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
