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
from . import _lib
from .dist import get_world_size, reduce_mean_many
from .models import BEVFusionCenterHead
from .ops import distill as D


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

    def train(self, mode=True):
        super().train(mode)
        self.teacher_model.train(mode and self.teacher_train_mode)
        return self

    # The step is split into phases so that neither the network pass nor its backward contains a
    # collective or a host sync (needed for hipGraph capture, useful for eager too):
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

    @torch.no_grad()
    def teacher(self, batch, prep):
        return self.teacher_model(self._points(batch), batch.get("imgs"), batch.get("mats_dict"),
                                  prep["gt"], return_feature=True)

    def student_loss(self, batch, prep, teacher_out, join=None):
        e = self.exp
        norm = prep["norm"]
        nh = norm.numel() - 2
        ret, tb, feat_s, bev_s, resp_s, _ = self.model(
            self._points(batch), batch.get("imgs"), batch.get("mats_dict"), prep["gt"],
            targets=prep["targets"], loss_norm=list(norm[:nh].unbind(0)))
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

    def forward(self, batch):
        prep = self.reduce(self.prep(batch))
        gt = prep["gt"]
        if not (self.overlap_teacher and gt.is_cuda) or torch.cuda.is_current_stream_capturing():
            return self.student_loss(batch, prep, self.teacher(batch, prep))
        # The teacher and the student forward are independent until the distillation losses, and both
        # are chains of short dependent kernels: on two streams the GPU fills one chain's gaps with the
        # other's kernels.  The teacher gets its own scratch buffers (ops on different streams must not
        # share a workspace) and its outputs are handed to the main stream explicitly.
        cur = torch.cuda.current_stream(gt.device)
        side = getattr(self, "_teacher_stream", None)
        if side is None or side.device != gt.device:
            side = self._teacher_stream = torch.cuda.Stream(gt.device)
        side.wait_stream(cur)
        with torch.cuda.stream(side), _lib.workspace_scope("teacher_stream"):
            tout = self.teacher(batch, prep)
        for part in tout:
            for t in _tensors_in(part):
                t.record_stream(cur)
        return self.student_loss(batch, prep, tout, join=side)


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
    """AdamW + grad clip + (optional) DDP around a module whose forward(batch) returns {'loss'}."""

    def __init__(self, step_module, lr=2e-4, weight_decay=1e-7, grad_clip=0.1, device=None,
                 bucket_cap_mb=64, autocast_dtype=None, channels_last=False):
        self.device = device or torch.device("cuda", torch.cuda.current_device())
        self.module = step_module.to(self.device)
        if channels_last:
            to_channels_last(self.module)
        self.module.train()
        trainable = [p for p in self.module.parameters() if p.requires_grad]
        self.ddp = None
        if get_world_size() > 1:
            # gradients are all-reduced over RCCL/xGMI in ~64 MB buckets, overlapped with backward
            self.ddp = nn.parallel.DistributedDataParallel(
                self.module, device_ids=[self.device.index], output_device=self.device.index,
                bucket_cap_mb=bucket_cap_mb,
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


def _tree_map(fn, x):
    if torch.is_tensor(x):
        return fn(x)
    if isinstance(x, dict):
        return {k: _tree_map(fn, v) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return type(x)(_tree_map(fn, v) for v in x)
    return x


def _tree_copy_(dst, src):
    if torch.is_tensor(dst):
        dst.copy_(src)
    elif isinstance(dst, dict):
        for k in dst:
            _tree_copy_(dst[k], src[k])
    elif isinstance(dst, (list, tuple)):
        for d, s_ in zip(dst, src):
            _tree_copy_(d, s_)


class GraphTrainer:
    """hipGraph-captured distillation trainer for students with static shapes (camera students).

    One eager step issues ~5 000 kernel launches and is bound by host dispatch; here the whole
    student side is captured once into three hipGraphs and replayed:
        G_prep     FCOS targets, box corners, gaussian mask, local loss normalisers
        [eager]    one all-reduce of the packed normalisers            (world > 1 only)
        [eager]    teacher sparse LiDAR encoder (dynamic shapes)  ->  G_tdense: teacher trunk + head
        G_student  zero grads, student forward, all losses, backward into ONE flat gradient buffer
        [eager]    all-reduce of the flat gradient buffer over RCCL    (world > 1 only)
        G_opt      grad-norm clip + fused AdamW
    No collective and no host sync sits inside a captured region, so the same graphs serve any
    world size.  Inputs are copied into static buffers before the replays.
    """

    def __init__(self, step_module, example_batch, lr=2e-4, weight_decay=1e-7, grad_clip=0.1,
                 device=None, autocast_dtype=None, warmup=3, channels_last=False):
        assert isinstance(step_module, DistillStep)
        self.device = device or torch.device("cuda", torch.cuda.current_device())
        self.m = step_module.to(self.device)
        if channels_last:
            to_channels_last(self.m)
        self.m.train()
        self.world = get_world_size()
        self.grad_clip = grad_clip
        self.ac = autocast_dtype
        self.params = [p for p in self.m.model.parameters() if p.requires_grad]
        # one flat fp32 gradient buffer; every .grad is a view into it (single RCCL all-reduce)
        n = sum(p.numel() for p in self.params)
        self.flat_grad = torch.zeros(n, dtype=torch.float32, device=self.device)
        off = 0
        for p in self.params:
            p.grad = self.flat_grad[off:off + p.numel()].view_as(p)
            off += p.numel()
        self.opt = torch.optim.AdamW(self.params, lr=lr, weight_decay=weight_decay, capturable=True,
                                     foreach=True)
        self.batch = _tree_map(lambda t: t.clone(), example_batch)      # static input buffers
        self.lidar_teacher = self.m.teacher_model.lidar_encoder is not None
        with _lib.workspace_scope(f"graph{id(self)}"):     # private scratch: pointers get baked in
            self._capture(warmup)

    # ---- pieces ---------------------------------------------------------------------------
    def _autocast(self):
        return torch.autocast("cuda", dtype=self.ac) if self.ac is not None else torch.autocast("cuda", enabled=False)

    def _teacher_sparse(self):
        """eager part of the teacher: sparse LiDAR encoder (host reads of voxel counts inside)."""
        t = self.m.teacher_model
        with torch.no_grad():
            return t.lidar_encoder(DistillStep._points(self.batch))

    def _teacher_dense(self, lidar_bev):
        t = self.m.teacher_model
        with torch.no_grad(), self._autocast():
            cam = t.camera_encoder(self.batch["imgs"], self.batch["mats_dict"]) if t.camera_encoder is not None else None
            if t.fusion_encoder is not None:
                bev = t.fusion_encoder(lidar_bev, cam)
            else:
                bev = cam if cam is not None else lidar_bev
            trunk, _ = t.bev_encoder(bev)
            ret = t.det_head(trunk, None)
        return bev, trunk, ret["multi_head_features"]

    def _student(self):
        self.flat_grad.zero_()
        with self._autocast():
            out = self.m.student_loss(self.batch, self.prep, self.teacher_out)
        out["loss"].backward()
        return out

    def _optimizer(self):
        if self.grad_clip:
            torch.nn.utils.clip_grad_norm_(self.params, self.grad_clip, foreach=True)
        self.opt.step()

    def _reduce_norm(self):
        if self.world > 1:
            torch.distributed.all_reduce(self.prep["local"], op=torch.distributed.ReduceOp.SUM)
            self.prep["local"].div_(float(self.world))
        self.prep["norm"] = self.prep["local"]

    # ---- capture --------------------------------------------------------------------------
    def _capture(self, warmup):
        # Warm-up on a side stream initialises MIOpen / hipBLASLt handles, workspaces and the
        # allocator (none of that is legal inside a capture); the model / optimizer state it
        # touched is restored afterwards so training starts from the caller's weights.
        warmup = max(int(warmup), 1)
        snap_m = {k: v.detach().clone() for k, v in self.m.state_dict().items()}
        side = torch.cuda.Stream(self.device)
        side.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(side):
            for _ in range(warmup):
                self.prep = self.m.prep(self.batch)
                self._reduce_norm()
                lb = self._teacher_sparse() if self.lidar_teacher else None
                self.teacher_out = self._teacher_dense(lb)
                self._student()
                self._optimizer()
        torch.cuda.current_stream(self.device).wait_stream(side)
        torch.cuda.synchronize(self.device)
        with torch.no_grad():
            self.m.load_state_dict(snap_m)
            for st in self.opt.state.values():          # exp_avg / exp_avg_sq / step back to zero
                for v in st.values():
                    if torch.is_tensor(v):
                        v.zero_()
        del snap_m
        self.g_prep = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.g_prep):
            self.prep = self.m.prep(self.batch)
        pool = self.g_prep.pool()
        self._reduce_norm()
        self.lidar_bev = self._teacher_sparse().clone() if self.lidar_teacher else None
        self.g_tdense = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.g_tdense, pool=pool):
            self.teacher_out = self._teacher_dense(self.lidar_bev)
        self.g_student = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.g_student, pool=pool):
            self.out = self._student()
        self.g_opt = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.g_opt, pool=pool):
            self._optimizer()
        torch.cuda.synchronize(self.device)

    # ---- one training step ----------------------------------------------------------------
    def step(self, batch):
        # Back-to-back replays without a host-side join fault on this ROCm ("write access to a read-only
        # page" inside a replayed graph); one stream join per step avoids it (tools/dbg_graph_b4.py).
        torch.cuda.current_stream(self.device).synchronize()
        if batch is not self.batch:
            _tree_copy_(self.batch, batch)
        self.g_prep.replay()
        self._reduce_norm()
        if self.lidar_teacher:
            self.lidar_bev.copy_(self._teacher_sparse())
        self.g_tdense.replay()
        self.g_student.replay()
        if self.world > 1:
            torch.distributed.all_reduce(self.flat_grad, op=torch.distributed.ReduceOp.SUM)
            self.flat_grad.div_(float(self.world))
        self.g_opt.replay()
        return self.out
