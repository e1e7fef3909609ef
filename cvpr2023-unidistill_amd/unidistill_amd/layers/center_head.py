"""CenterPoint-style IoU-aware detection head, FCOS target assigner and the detection losses.

State_dict-compatible mirrors of
  CenterHead / SepHead            unidistill/layers/head/det3d/center_head.py:14-146, :311-375
  CenterHeadIouAware              unidistill/layers/head/det3d/center_head_iou_aware.py:12-298
  FCOSAssigner                    unidistill/layers/head/det3d/target_assigner/fcos_assigner.py:9-285
  FocalLoss / CenterNetRegLoss / AutomaticWeightedLoss
                                  unidistill/layers/losses/det3d.py:10-34, :287-319, :382-421
  boxes3d_nearest_bev_iou         unidistill/utils/det3d_utils/box_utils.py:343-373
Same numbers, different execution: the reference walks python loops over samples x tasks with
boolean-mask indexing (dynamic shapes -> a device sync per step of the loop), ``.item()`` on ~15
scalars per task and a python ``if`` on ``loc_loss.item() < 1``.  Everything here is static-shape
tensor code with device-side selects: no host synchronisation anywhere in forward or loss.
The dense convs run through PyTorch-ROCm (MIOpen).
"""
import copy
import math

import torch
from torch import nn

from .. import _lib
from ..dist import reduce_mean, reduce_mean_many
from ..ops import bn_act as hipbn, conv2d as hipconv, conv2d_f32 as hipconv32, det_loss as hiploss, head_tail, \
    head_tail_f32
from .dense import Conv2d, FusedSequential


# --------------------------------------------------------------------------------------------
# losses
# --------------------------------------------------------------------------------------------
class AutomaticWeightedLoss(nn.Module):
    """sum_i 0.5 / p_i^2 * loss_i + log(1 + p_i^2)  with learnable p (init 1)."""

    def __init__(self, num=2):
        super().__init__()
        self.params = nn.Parameter(torch.ones(num))

    def forward(self, *losses):
        total = 0
        for i, l in enumerate(losses):
            p2 = self.params[i] ** 2
            total = total + 0.5 / p2 * l + torch.log(1 + p2)
        return total


class FocalLoss(nn.Module):
    """CornerNet focal loss on post-sigmoid heat maps (losses/det3d.py:287-319)."""

    def __init__(self, alpha, gamma):
        super().__init__()
        self.alpha, self.gamma = alpha, gamma

    def forward(self, pred, gt, num_pos=None):
        pos = gt.eq(1)
        neg = gt.eq(0)
        pos_loss = (torch.log(pred) * torch.pow(1 - pred, self.gamma) * pos.long() * self.alpha).sum()
        neg_loss = (torch.log(1 - pred + 1e-4) * torch.pow(pred, self.gamma) * neg.long()
                    * (1 - self.alpha)).sum()
        if num_pos is None:
            num_pos = reduce_mean(pos.float().sum())
        # reference: ``if num_pos == 0: -neg else -(pos + neg) / num_pos`` (a host-side branch)
        safe = torch.where(num_pos == 0, torch.ones_like(num_pos), num_pos)
        return torch.where(num_pos == 0, -neg_loss, -(pos_loss + neg_loss) / safe)


def _transpose_and_gather_feat(feat, ind):
    """feat [B,C,H,W], ind i64[B,K] (flattened y*W+x) -> [B,K,C]."""
    b, c = feat.shape[:2]
    flat = feat.reshape(b, c, -1)
    return flat.gather(2, ind.unsqueeze(1).expand(b, c, ind.shape[1])).transpose(1, 2).contiguous()


class CenterNetRegLoss(nn.Module):
    """Masked L1 per code dimension, normalised by the (global mean) number of objects."""

    def forward(self, output, mask, ind, target, num=None):
        pred = _transpose_and_gather_feat(output, ind)
        if num is None:
            num = reduce_mean(mask.float().sum())
        m = mask.unsqueeze(2).float() * (~torch.isnan(target)).float()
        loss = torch.abs(pred * m - target * m)          # [B,K,dim]
        return loss.sum(dim=(0, 1)) / (num + 1e-4)


def limit_period(val, offset=0.5, period=math.pi):
    return val - torch.floor(val / period + offset) * period


def nearest_bev_iou_pairwise(boxes_a, boxes_b):
    """Diagonal of boxes3d_nearest_bev_iou(a, b): IoU of the axis-aligned BEV boxes of a[i], b[i].
    (The reference builds the full N x N matrix, N = B*2500, and keeps the diagonal.)"""
    def aligned(b):
        rot = limit_period(b[:, 6], 0.5, math.pi).abs()
        dims = torch.where(rot[:, None] < math.pi / 4, b[:, 3:5], b[:, 3:5].flip(1))   # no host index lists (graph capture)
        return b[:, 0:2] - dims / 2, b[:, 0:2] + dims / 2
    a0, a1 = aligned(boxes_a)
    b0, b1 = aligned(boxes_b)
    inter = torch.clamp_min(torch.min(a1, b1) - torch.max(a0, b0), 0)
    inter = inter[:, 0] * inter[:, 1]
    area_a = (a1[:, 0] - a0[:, 0]) * (a1[:, 1] - a0[:, 1])
    area_b = (b1[:, 0] - b0[:, 0]) * (b1[:, 1] - b0[:, 1])
    return inter / torch.clamp_min(area_a + area_b - inter, 1e-6)


def boxes3d_nearest_bev_iou(boxes_a, boxes_b):
    """Full [N,M] matrix variant (box_utils.py:361-373), for API parity."""
    n, m = boxes_a.shape[0], boxes_b.shape[0]
    a = boxes_a[:, None, :].expand(n, m, boxes_a.shape[1]).reshape(n * m, -1)
    b = boxes_b[None, :, :].expand(n, m, boxes_b.shape[1]).reshape(n * m, -1)
    return nearest_bev_iou_pairwise(a, b).reshape(n, m)


# --------------------------------------------------------------------------------------------
# target assignment
# --------------------------------------------------------------------------------------------
class FCOSAssigner:
    """Top-k nearest anchor points per GT box, per task (fcos_assigner.py:73-285), batched."""

    def __init__(self, out_size_factor, tasks, dense_reg, gaussian_overlap, max_objs, min_radius,
                 mapping, grid_size, pc_range, voxel_size, assign_topk, no_log=False,
                 with_velocity=False):
        self.out_size_factor = out_size_factor
        self.tasks = tasks
        self.task_classes = [list(t["class_names"]) for t in tasks]
        self.num_classes = sum(len(c) for c in self.task_classes)
        self.dense_reg = dense_reg
        self._max_objs = max_objs
        self.class_to_idx = mapping
        self.grid_size = [int(g) for g in grid_size]
        self.pc_range = pc_range
        self.voxel_size = voxel_size
        self.assign_topk = assign_topk
        self.with_velocity = with_velocity
        self.default_box_dims = 10 if with_velocity else 8
        self._anchors = {}

    def anchor_points(self, device):
        a = self._anchors.get(device)
        if a is None:
            w, h = self.grid_size[0] // self.out_size_factor, self.grid_size[1] // self.out_size_factor
            s = self.out_size_factor
            gx = torch.linspace(0.0, (w - 1) * s, steps=w, device=device)
            gy = torch.linspace(0.0, (h - 1) * s, steps=h, device=device)
            a = torch.stack([gx.repeat(h), gy.repeat_interleave(w)], 1)    # [H*W, 2], x fastest
            self._anchors[device] = a
        return a

    def _class_tables(self, device):
        tabs = getattr(self, "_tabs", {}).get(device)
        if tabs is None:
            ncls = max(self.class_to_idx.values()) + 1
            task_of = torch.full((ncls,), -1, dtype=torch.long)
            off_of = torch.zeros((ncls,), dtype=torch.long)
            for t, names in enumerate(self.task_classes):
                for o, name in enumerate(names):
                    task_of[self.class_to_idx[name]] = t
                    off_of[self.class_to_idx[name]] = o
            tabs = (task_of.to(device), off_of.to(device))
            if not hasattr(self, "_tabs"):
                self._tabs = {}
            self._tabs[device] = tabs
        return tabs

    fused = True        # one HIP kernel per step on the GPU (csrc/assign.hip); tensor ops otherwise

    def _assign_hip(self, gt_boxes, T, B, M, K, w, h):
        """ud_assign_targets: the whole assignment in one launch.  Returns None when the configuration is
        outside the kernel's limits (the tensor-op path then runs)."""
        import ctypes
        from .. import _lib
        cols = gt_boxes.shape[2]
        ncls = max(self.class_to_idx.values()) + 1
        topk = min(self.assign_topk, w * h)
        if M > 512 or cols > 12 or cols < 8 or ncls > 64 or topk > 9 or w < 9 or h < 9 or w * h > 180 * 180:
            return None
        tabs = getattr(self, "_host_tabs", None)
        if tabs is None:
            task_of, off_of = [-1] * ncls, [0] * ncls
            for t, names in enumerate(self.task_classes):
                for o, name in enumerate(names):
                    task_of[self.class_to_idx[name]] = t
                    off_of[self.class_to_idx[name]] = o
            tabs = self._host_tabs = ((ctypes.c_byte * ncls)(*task_of), (ctypes.c_byte * ncls)(*off_of))
        ncmax = max(len(c) for c in self.task_classes)
        enc_dim = max(8 + max(cols - 1 - 7, 0), self.default_box_dims)
        dev = gt_boxes.device
        gt = gt_boxes.contiguous()
        hm = torch.empty((T, B, ncmax, h, w), dtype=torch.float32, device=dev)
        ind = torch.empty((T, B, K), dtype=torch.long, device=dev)
        mask = torch.empty((T, B, K), dtype=torch.bool, device=dev)
        cat = torch.empty((T, B, K), dtype=torch.long, device=dev)
        enc = torch.empty((T, B, K, enc_dim), dtype=torch.float32, device=dev)
        lib = _lib.load()
        _lib.check(lib.ud_assign_targets(_lib.ptr(gt), B, M, cols, tabs[0], tabs[1], ncls, T, ncmax, w, h, K, topk,
                                         enc_dim, float(self.out_size_factor), float(self.pc_range[0]),
                                         float(self.pc_range[1]), float(self.voxel_size[0]),
                                         float(self.voxel_size[1]), _lib.ptr(hm), _lib.ptr(ind), _lib.ptr(mask),
                                         _lib.ptr(cat), _lib.ptr(enc), _lib.stream_of(gt)), "ud_assign_targets")
        out = {k: {} for k in ("heatmap", "ind", "mask", "cat", "box_encoding")}
        for t, names in enumerate(self.task_classes):
            out["heatmap"][t] = hm[t, :, :len(names)].contiguous() if len(names) != ncmax else hm[t]
            out["ind"][t], out["mask"][t], out["cat"][t] = ind[t], mask[t], cat[t]
            out["box_encoding"][t] = enc[t]
        out["_stacked"] = {"heatmap": hm, "ind": ind, "mask": mask, "box_encoding": enc}
        return out

    windowed = True     # candidate anchors from a 9x9 window around each box centre (see _assign_windowed)

    def _assign_dense(self, anchors, pcx, pcy, mem, K, topk):
        """Reference formulation: all [TB, M, A] squared centre distances, top-k per box, argmin per anchor."""
        dev = anchors.device
        TB, M = pcx.shape
        A = anchors.shape[0]
        d = (anchors[None, None, :, 0] - pcx[:, :, None]) ** 2 + (anchors[None, None, :, 1] - pcy[:, :, None]) ** 2
        d = torch.where(mem[:, :, None], d, torch.full_like(d, float("inf")))
        tk = torch.topk(d, topk, dim=2, largest=False).indices                           # [TB,M,topk]
        hits = torch.zeros((TB, A), device=dev)
        hits.scatter_add_(1, tk.reshape(TB, -1), mem[:, :, None].expand(TB, M, topk).reshape(TB, -1).float())
        pos = hits > 0                                                                   # [TB,A]
        gid = d.argmin(1)                                                                # nearest GT (task order)
        rank = pos.long().cumsum(1) - 1
        slot = torch.where(pos & (rank < K), rank, torch.full_like(rank, K))
        aidx = torch.arange(A, device=dev)[None, :].expand(TB, A)
        ind = torch.zeros((TB, K + 1), dtype=torch.long, device=dev).scatter_(1, slot, aidx)[:, :K]
        sgt = torch.zeros((TB, K + 1), dtype=torch.long, device=dev).scatter_(1, slot, gid)[:, :K]
        mask = torch.zeros((TB, K + 1), dtype=torch.bool, device=dev).scatter_(1, slot, pos)[:, :K]
        return ind, sgt, mask, pos, gid

    def _assign_windowed(self, anchors, pcx, pcy, mem, w, h, K, topk):
        """Same result without the [TB, M, A] distance tensor (124 MB at B=4): the k <= 9 anchors nearest
        to a point lie within +-4 cells of the (clamped) nearest cell of the regular anchor grid, so the
        top-k runs over 81 candidates per box; positives are the <= M*k selected anchors, sorted and
        de-duplicated; the nearest box is searched only at those positives."""
        dev = anchors.device
        TB, M = pcx.shape
        s = float(self.out_size_factor)
        inf = float("inf")
        cxi = torch.round(pcx / s).clamp(0, w - 1).long()
        cyi = torch.round(pcy / s).clamp(0, h - 1).long()
        off = torch.arange(-4, 5, device=dev)
        ix = cxi[:, :, None, None] + off[None, None, None, :]                            # [TB,M,1,9]
        iy = cyi[:, :, None, None] + off[None, None, :, None]                            # [TB,M,9,1]
        ok = (ix >= 0) & (ix < w) & (iy >= 0) & (iy < h) & mem[:, :, None, None]
        d = (ix.float() * s - pcx[:, :, None, None]) ** 2 + (iy.float() * s - pcy[:, :, None, None]) ** 2
        d = torch.where(ok, d, torch.full_like(d, inf)).reshape(TB, M, 81)
        aid = (iy * w + ix).reshape(TB, M, 81)
        dk, sel = torch.topk(d, topk, dim=2, largest=False)
        cand = aid.gather(2, sel).reshape(TB, M * topk)                                  # anchor ids
        cok = torch.isfinite(dk).reshape(TB, M * topk)
        A = w * h
        key = torch.where(cok, cand, torch.full_like(cand, A))
        key, _ = torch.sort(key, dim=1)                                                  # ascending anchors
        first = torch.ones_like(key, dtype=torch.bool)
        first[:, 1:] = key[:, 1:] != key[:, :-1]
        pos_ok = first & (key < A)                                                       # unique positives
        pos_ids = torch.where(pos_ok, key, torch.zeros_like(key))
        # nearest box (task order, first on ties) at every positive anchor
        ax = (pos_ids % w).float() * s
        ay = (pos_ids // w).float() * s
        dg = (ax[:, None, :] - pcx[:, :, None]) ** 2 + (ay[:, None, :] - pcy[:, :, None]) ** 2   # [TB,M,P]
        dg = torch.where(mem[:, :, None], dg, torch.full_like(dg, inf))
        pos_gid = dg.argmin(1)                                                           # [TB,P]
        rank = pos_ok.long().cumsum(1) - 1
        slot = torch.where(pos_ok & (rank < K), rank, torch.full_like(rank, K))
        ind = torch.zeros((TB, K + 1), dtype=torch.long, device=dev).scatter_(1, slot, pos_ids)[:, :K]
        sgt = torch.zeros((TB, K + 1), dtype=torch.long, device=dev).scatter_(1, slot, pos_gid)[:, :K]
        mask = torch.zeros((TB, K + 1), dtype=torch.bool, device=dev).scatter_(1, slot, pos_ok)[:, :K]
        return ind, sgt, mask, pos_ids, pos_gid, pos_ok

    @torch.no_grad()
    def assign_targets(self, gt_boxes):
        """gt_boxes f32[B,M,C+1] (last column = 1-based class) -> dict of per-task targets.
        All tasks are processed at once: the task index is folded into the batch dimension
        (TB = T*B "samples"), so the op count does not grow with the number of tasks."""
        dev = gt_boxes.device
        B, M = gt_boxes.shape[:2]
        T = len(self.task_classes)
        K = self._max_objs * self.dense_reg
        w, h = self.grid_size[0] // self.out_size_factor, self.grid_size[1] // self.out_size_factor
        if self.fused and gt_boxes.is_cuda and gt_boxes.dtype == torch.float32:
            out = self._assign_hip(gt_boxes, T, B, M, K, w, h)
            if out is not None:
                return out
        anchors = self.anchor_points(dev)
        A = anchors.shape[0]
        ncmax = max(len(c) for c in self.task_classes)
        task_of, off_of = self._class_tables(dev)
        cls = gt_boxes[:, :, -1].long().clamp(0, task_of.numel() - 1)
        box = gt_boxes[:, :, :-1]
        idx = torch.arange(M, device=dev)
        # rows up to the last one that does not sum to zero are kept (row 0 always is)
        nz = box.sum(-1) != 0
        last = torch.where(nz, idx, torch.zeros_like(idx)).amax(1, keepdim=True)
        valid = idx[None, :] <= last
        vs0, vs1 = self.voxel_size[0], self.voxel_size[1]
        cx = (box[:, :, 0] - self.pc_range[0]) / vs0
        cy = (box[:, :, 1] - self.pc_range[1]) / vs1
        dxv, dyv = box[:, :, 3] / vs0, box[:, :, 4] / vs1
        yaw = limit_period(box[:, :, 6], offset=0.5, period=math.pi * 2)
        # class offset inside each task, -1 for boxes of other tasks / invalid rows: [T,B,M] -> [TB,M]
        tsel = torch.arange(T, device=dev)[:, None, None]
        member = valid[None] & (task_of[cls][None] == tsel)
        coff = torch.where(member, off_of[cls][None].expand(T, B, M), torch.full((T, B, M), -1, device=dev))
        coff = coff.reshape(T * B, M)
        member = member.reshape(T * B, M)
        rep = lambda v: v[None].expand(T, *v.shape).reshape(T * B, *v.shape[1:])
        TB = T * B
        # task order = (class offset, original index): order matters only for argmin ties
        key = torch.where(member, coff * M + idx[None, :], torch.full_like(coff, ncmax * M + M))
        perm = key.argsort(1)
        mem = member.gather(1, perm)
        pcx, pcy = rep(cx).gather(1, perm), rep(cy).gather(1, perm)
        topk = min(self.assign_topk, A)
        if self.windowed and topk <= 9 and w >= 9 and h >= 9:
            ind, sgt, mask, pos_ids, pos_gid, pos_ok = self._assign_windowed(anchors, pcx, pcy, mem, w, h, K, topk)
            cat_src = (pos_ids, pos_gid, pos_ok)
        else:
            ind, sgt, mask, pos, gid = self._assign_dense(anchors, pcx, pcy, mem, K, topk)
            cat_src = None
        sperm = perm.gather(1, sgt)                                                      # original box row
        cat = torch.where(mask, coff.gather(1, sperm), torch.zeros_like(sgt))
        # heat map: one-hot of the assigned class at every positive anchor
        if cat_src is None:
            hm = torch.zeros((TB, ncmax, A), device=dev)
            cat_anchor = coff.gather(1, perm.gather(1, gid)).clamp_min(0)
            hm.scatter_(1, cat_anchor[:, None, :], pos[:, None, :].float())
        else:
            pos_ids, pos_gid, pos_ok = cat_src                                         # [TB, M*topk]
            cls_of = coff.gather(1, perm.gather(1, pos_gid)).clamp_min(0)
            flat = cls_of * A + pos_ids                                                # index into [ncmax*A]
            # invalid candidates write a 0 at (class 0, anchor 0 ...): harmless only if nothing valid
            # writes there too, so route them to a scratch column instead
            hm_flat = torch.zeros((TB, ncmax * A + 1), device=dev)
            hm_flat.scatter_(1, torch.where(pos_ok, flat, torch.full_like(flat, ncmax * A)),
                             pos_ok.float())
            hm = hm_flat[:, :ncmax * A].reshape(TB, ncmax, A)
        # box encoding of the assigned GT relative to its anchor point
        g = lambda v: rep(v).gather(1, sperm)
        ax, ay = anchors[:, 0][ind], anchors[:, 1][ind]
        cols = [(g(cx) - ax) / self.out_size_factor, (g(cy) - ay) / self.out_size_factor,
                g(box[:, :, 2]), torch.log(g(dxv) * vs0), torch.log(g(dyv) * vs1),
                torch.log(g(box[:, :, 5])), torch.sin(g(yaw)), torch.cos(g(yaw))]
        cols += [g(box[:, :, j]) for j in range(7, box.shape[2])]
        enc = torch.stack(cols, 2)
        enc = torch.where(mask[:, :, None], enc, torch.zeros_like(enc))                 # padded slots = 0
        if enc.shape[2] < self.default_box_dims:
            enc = torch.cat([enc, enc.new_zeros(TB, K, self.default_box_dims - enc.shape[2])], 2)
        ind = torch.where(mask, ind, torch.zeros_like(ind))
        hm = hm.reshape(T, B, ncmax, h, w)
        ind, mask, cat, enc = (v.reshape(T, B, *v.shape[1:]) for v in (ind, mask, cat, enc.float()))
        out = {k: {} for k in ("heatmap", "ind", "mask", "cat", "box_encoding")}
        for t, names in enumerate(self.task_classes):
            out["heatmap"][t] = hm[t, :, :len(names)].contiguous()
            out["ind"][t], out["mask"][t], out["cat"][t] = ind[t], mask[t], cat[t]
            out["box_encoding"][t] = enc[t]
        # stacked views for the task-vectorised loss
        out["_stacked"] = {"heatmap": hm, "ind": ind, "mask": mask, "box_encoding": enc}
        return out


# --------------------------------------------------------------------------------------------
# head
# --------------------------------------------------------------------------------------------
class _HeadSeq(nn.Sequential):
    """nn.Sequential with the ``add`` of the reference's tiny Sequential (center_head.py:260-308)."""

    def add(self, module, name=None):
        self.add_module(str(len(self._modules)) if name is None else name, module)


class SepHead(nn.Module):
    """One small conv stack per output quantity (center_head.py:311-375)."""

    def __init__(self, in_channels, heads, head_conv=64, final_kernel=1, bn=False, init_bias=-2.19,
                 directional_classifier=False, **_):
        super().__init__()
        assert directional_classifier is False
        self.heads = heads
        for head, (classes, num_conv) in heads.items():
            fc = _HeadSeq()
            for _i in range(num_conv - 1):
                fc.add(nn.Conv2d(in_channels, head_conv, final_kernel, padding=final_kernel // 2, bias=True))
                if bn:
                    fc.add(nn.BatchNorm2d(head_conv))
                fc.add(nn.ReLU())
            fc.add(nn.Conv2d(head_conv, classes, final_kernel, padding=final_kernel // 2, bias=True))
            if "hm" in head:
                fc[-1].bias.data.fill_(init_bias)
            else:
                for m in fc.modules():
                    if isinstance(m, nn.Conv2d):
                        nn.init.kaiming_normal_(m.weight, a=0, mode="fan_out", nonlinearity="relu")
                        nn.init.constant_(m.bias, 0)
            setattr(self, head, fc)

    def forward(self, x):
        return {head: getattr(self, head)(x) for head in self.heads}


class PackedSepHeads(nn.Module):
    """All SepHeads of all tasks as TWO convolutions.

    The reference runs 6 tasks x 7 heads = 42 independent (conv3x3 64->64, BN, ReLU, conv3x3 64->k)
    stacks: 84 tiny convolutions + 42 batch norms per pass (x3 with backward).  Every first conv
    reads the same shared feature map, so they are ONE conv 64 -> 42*64; the BatchNorms are one
    BatchNorm over 42*64 channels (per-channel statistics are unaffected by packing); the second
    convs are ONE grouped conv (42 groups, outputs padded to the widest head, unused rows stay 0).
    Parameters live packed; ``state_dict`` / ``load_state_dict`` speak the reference's key names
    (``{t}.{head}.0.weight`` ... relative to this module, i.e. ``tasks.{t}.{head}...`` in the head).
    """

    def __init__(self, in_channels, task_heads, head_conv=64, final_kernel=3, init_bias=-2.19):
        super().__init__()
        self.in_channels, self.head_conv, self.k = in_channels, head_conv, final_kernel
        self.layout = []                                   # (task, head name, out channels)
        for t, heads in enumerate(task_heads):
            for name, (classes, num_conv) in heads.items():
                assert num_conv == 2, "UniDistill's heads are (conv, BN, ReLU, conv)"
                self.layout.append((t, name, int(classes)))
        self.num_tasks = len(task_heads)
        G = len(self.layout)
        self.kmax = max(k for _, _, k in self.layout)
        hc = head_conv
        self.c1_weight = nn.Parameter(torch.empty(G * hc, in_channels, final_kernel, final_kernel))
        self.c1_bias = nn.Parameter(torch.empty(G * hc))
        self.bn_weight = nn.Parameter(torch.ones(G * hc))
        self.bn_bias = nn.Parameter(torch.zeros(G * hc))
        self.register_buffer("bn_running_mean", torch.zeros(G * hc))
        self.register_buffer("bn_running_var", torch.ones(G * hc))
        self.register_buffer("bn_num_batches_tracked", torch.tensor(0, dtype=torch.long))
        self.c2_weight = nn.Parameter(torch.zeros(G * self.kmax, hc, final_kernel, final_kernel))
        self.c2_bias = nn.Parameter(torch.zeros(G * self.kmax))
        self.bn_eps, self.bn_momentum = 1e-5, 0.1
        self.fused_tail = True      # HIP tail kernel when the hidden tensor is CUDA bf16
        self.fused_bn_tail_f32 = True   # fp32: BatchNorm + ReLU inside the grouped tail kernels (False: separate streaming pass)
        self.register_buffer("_c2_rows", torch.arange(G * self.kmax), persistent=False)
        self.register_buffer("_c2_grp", torch.arange(G * self.kmax) // self.kmax, persistent=False)
        # initialise exactly like the reference's per-head modules (center_head.py:323-362)
        with torch.no_grad():
            for g, (_t, name, kout) in enumerate(self.layout):
                c1 = nn.Conv2d(in_channels, hc, final_kernel, padding=final_kernel // 2, bias=True)
                c2 = nn.Conv2d(hc, kout, final_kernel, padding=final_kernel // 2, bias=True)
                if "hm" in name:
                    c2.bias.fill_(init_bias)
                else:
                    for m in (c1, c2):
                        nn.init.kaiming_normal_(m.weight, a=0, mode="fan_out", nonlinearity="relu")
                        nn.init.constant_(m.bias, 0)
                self.c1_weight[g * hc:(g + 1) * hc] = c1.weight
                self.c1_bias[g * hc:(g + 1) * hc] = c1.bias
                self.c2_weight[g * self.kmax:g * self.kmax + kout] = c2.weight
                self.c2_bias[g * self.kmax:g * self.kmax + kout] = c2.bias
        self._register_state_dict_hook(self._to_reference_keys)
        self._register_load_state_dict_pre_hook(self._from_reference_keys)

    def __len__(self):
        return self.num_tasks

    def forward(self, x):
        """-> list (per task) of {head: [B, k, H, W]} like [SepHead(x) for each task]."""
        pad = self.k // 2
        if Conv2d.hip_enabled and self.k == 3 and x.is_cuda and x.dtype == torch.bfloat16 \
                and hipconv.supported(x, self.c1_weight):
            y = hipconv.conv3x3(x, self.c1_weight, self.c1_bias)      # 64 -> 42*64 on the MFMA kernel
        elif Conv2d.hip_enabled and Conv2d.hip_fp32 and self.k == 3 and x.is_cuda and x.dtype == torch.float32 \
                and not torch.is_autocast_enabled("cuda") and hipconv32.supported(x, self.c1_weight, 3):
            # fp32 MFMA kernel (1.2x the library, deterministic); in training its epilogue also emits the per-tile sums of
            # the BatchNorm over the 2 688-channel hidden tensor (one 1.39 GB read less)
            y = hipconv32.conv3x3(x, self.c1_weight, self.c1_bias, self.training)
        else:
            if Conv2d.hip_enabled:
                _lib.library_fallthrough("layers.center_head.PackedSepHeads (first conv)", x, self.c1_weight)
            y = torch.nn.functional.conv2d(x, self.c1_weight, self.c1_bias, padding=pad)
        if self.training:
            self.bn_num_batches_tracked += 1
        G = len(self.layout)
        frozen_bn_grad = (not self.training) and torch.is_grad_enabled() and y.requires_grad
        if self.fused_tail and head_tail.supported(y, self.head_conv, self.kmax, self.k) and not frozen_bn_grad:
            # bf16 mixed-precision mode on the GPU: BN + ReLU + all 42 second convs in one HBM-bound
            # kernel (ops/head_tail.py); y is read once instead of 3 r/w passes + a 42x padded conv
            z = head_tail.head_tail(y, self.bn_weight, self.bn_bias, self.c2_weight, self.c2_bias,
                                    self.bn_running_mean, self.bn_running_var, self.training,
                                    self.bn_momentum, self.bn_eps, G, self.kmax)
            return self._split(z)
        if Conv2d.hip_enabled and Conv2d.hip_fp32 and not torch.is_autocast_enabled("cuda") and not frozen_bn_grad \
                and head_tail_f32.supported(y, self.head_conv, self.kmax, self.k) and (self.training or not torch.is_grad_enabled()):
            # fp32 mode on the GPU: BN + ReLU as one streaming pass, then the 42 second convs as ONE grouped fp32
            # kernel (the block-diagonal dense conv below costs 42x the FLOPs: 19 ms per fwd+bwd at B = 4)
            partial = getattr(y, "_ud_bn_partial", None) if self.training else None
            if self.fused_bn_tail_f32:
                # ... with BatchNorm + ReLU applied inside the tail kernels as they load y: the normalised hidden tensor never exists
                return self._split(head_tail_f32.bn_relu_group_tail(
                    y, self.bn_weight, self.bn_bias, self.bn_running_mean, self.bn_running_var, self.training,
                    self.bn_momentum, self.bn_eps, None, partial, self.c2_weight, self.c2_bias, G, self.kmax))
            a = hipbn._BnActFn.apply(y if y.is_contiguous(memory_format=torch.channels_last)
                                     else y.contiguous(memory_format=torch.channels_last),
                                     self.bn_weight, self.bn_bias, None, self.bn_running_mean, self.bn_running_var,
                                     self.training, self.bn_momentum, self.bn_eps, True, None, partial)
            return self._split(head_tail_f32.group_tail(a, self.c2_weight, self.c2_bias, G, self.kmax))
        if Conv2d.hip_enabled:
            _lib.library_fallthrough("layers.center_head.PackedSepHeads (BN + second convs)", y, self.c2_weight,
                                     training=self.training, grad=torch.is_grad_enabled())
        y = torch.nn.functional.batch_norm(y, self.bn_running_mean, self.bn_running_var, self.bn_weight,
                                           self.bn_bias, self.training, self.bn_momentum, self.bn_eps)
        y = torch.relu(y)
        # second layer: ONE dense conv with a block-diagonal weight built from the compact per-head
        # parameters (rows padded to a multiple of 64).  A 42-group conv with 3 outputs per group is
        # ~5x slower in MIOpen/CK than the dense conv despite 42x fewer FLOPs (tools/exp_gconv.py).
        rows = G * self.kmax
        rows_p = (rows + 63) // 64 * 64
        wd = self.c2_weight.new_zeros(rows_p, G, self.head_conv, self.k, self.k)
        wd = wd.index_put((self._c2_rows, self._c2_grp), self.c2_weight)
        bd = torch.nn.functional.pad(self.c2_bias, (0, rows_p - rows))
        z = torch.nn.functional.conv2d(y, wd.view(rows_p, G * self.head_conv, self.k, self.k), bd, padding=pad)
        return self._split(z)

    def _split(self, z):
        # ONE split node: its backward is a single concatenation of the 42 head gradients, where 42
        # independent slices would each zero-pad to the full [B, 126, H, W] tensor and add (1.5 ms/step)
        sizes = []
        for _t, _name, kout in self.layout:
            sizes.append(kout)
            if self.kmax > kout:
                sizes.append(self.kmax - kout)
        rest = z.shape[1] - sum(sizes)
        if rest:
            sizes.append(rest)
        if self.training and z.is_cuda and z.dtype in (torch.float32, torch.bfloat16) \
                and not (z.is_contiguous() and z.dtype == torch.float32):
            # channels-last (and, under autocast, bf16) tail output of the training head: ONE copy to fp32 planes, so that
            # the head slices are the contiguous [H, W] planes the fused detection-loss kernels read (ops/det_loss.py: 2
            # launches) -- on strided slices get_loss falls back to its tensor-op formulation, ~330 small launches forward
            # and as many backward
            z = z.to(torch.float32, memory_format=torch.contiguous_format)
        parts = iter(z.split_with_sizes(sizes, dim=1))
        outs = [dict() for _ in range(self.num_tasks)]
        for t, name, kout in self.layout:
            outs[t][name] = next(parts)
            if self.kmax > kout:
                next(parts)
        return outs

    # ---- reference-compatible (de)serialisation ------------------------------------------------
    def _slices(self):
        hc, km = self.head_conv, self.kmax
        for g, (t, name, kout) in enumerate(self.layout):
            yield f"{t}.{name}.", slice(g * hc, (g + 1) * hc), slice(g * km, g * km + kout)

    def _to_reference_keys(self, module, state, prefix, local_metadata):
        packed = {k: state.pop(prefix + k) for k in ("c1_weight", "c1_bias", "bn_weight", "bn_bias",
                                                      "bn_running_mean", "bn_running_var",
                                                      "bn_num_batches_tracked", "c2_weight", "c2_bias")}
        for key, s1, s2 in self._slices():
            state[prefix + key + "0.weight"] = packed["c1_weight"][s1]
            state[prefix + key + "0.bias"] = packed["c1_bias"][s1]
            state[prefix + key + "1.weight"] = packed["bn_weight"][s1]
            state[prefix + key + "1.bias"] = packed["bn_bias"][s1]
            state[prefix + key + "1.running_mean"] = packed["bn_running_mean"][s1]
            state[prefix + key + "1.running_var"] = packed["bn_running_var"][s1]
            state[prefix + key + "1.num_batches_tracked"] = packed["bn_num_batches_tracked"].clone()
            state[prefix + key + "3.weight"] = packed["c2_weight"][s2]
            state[prefix + key + "3.bias"] = packed["c2_bias"][s2]
        return state

    def _from_reference_keys(self, state, prefix, local_metadata, strict, missing, unexpected, errors):
        if prefix + "c1_weight" in state:
            return                                          # already packed
        first = prefix + self.layout[0][1].join([f"{self.layout[0][0]}.", ".0.weight"])
        if first not in state:
            return                                          # nothing for us: let strict mode report it
        ref = {k: state.pop(k) for k in list(state) if k.startswith(prefix)}
        like = lambda p: torch.zeros_like(p.detach())
        new = {"c1_weight": like(self.c1_weight), "c1_bias": like(self.c1_bias),
               "bn_weight": like(self.bn_weight), "bn_bias": like(self.bn_bias),
               "bn_running_mean": torch.zeros_like(self.bn_running_mean),
               "bn_running_var": torch.ones_like(self.bn_running_var),
               "bn_num_batches_tracked": self.bn_num_batches_tracked.detach().clone(),
               "c2_weight": like(self.c2_weight), "c2_bias": like(self.c2_bias)}
        names = {"0.weight": ("c1_weight", 1), "0.bias": ("c1_bias", 1), "1.weight": ("bn_weight", 1),
                 "1.bias": ("bn_bias", 1), "1.running_mean": ("bn_running_mean", 1),
                 "1.running_var": ("bn_running_var", 1), "3.weight": ("c2_weight", 2), "3.bias": ("c2_bias", 2)}
        for key, s1, s2 in self._slices():
            for suffix, (dst, which) in names.items():
                k = prefix + key + suffix
                if k in ref:
                    new[dst][s1 if which == 1 else s2] = ref.pop(k).to(new[dst].dtype)
                elif strict:
                    missing.append(k)
            k = prefix + key + "1.num_batches_tracked"
            if k in ref:
                new["bn_num_batches_tracked"] = ref.pop(k).to(torch.long).reshape(())
        for k, v in new.items():
            state[prefix + k] = v
        state.update(ref)                                   # leftovers become "unexpected" in strict mode


class CenterHead(nn.Module):
    def __init__(self, dataset_name, tasks, target_assigner, proposal_layer, input_channels,
                 grid_size, point_cloud_range, code_weights, loc_weight, share_conv_channel,
                 common_heads, upsample_for_pedestrian=False, predict_boxes_when_training=False,
                 mode="3d", init_bias=-2.19, distill=False, packed_heads=True):
        super().__init__()
        self.in_channels = input_channels
        self.grid_size = grid_size
        self.point_cloud_range = point_cloud_range
        self.predict_boxes_when_training = predict_boxes_when_training
        self.num_classes = [len(t["class_names"]) for t in tasks]
        self.class_names = [t["class_names"] for t in tasks]
        self.code_weights = code_weights
        self.weight = loc_weight
        self.dataset = dataset_name
        self.box_n_dim = 9 if dataset_name == "nuscenes" else 7
        self.distill = distill
        self.shared_conv = FusedSequential(
            Conv2d(input_channels, share_conv_channel, 3, padding=1, bias=True),
            nn.BatchNorm2d(share_conv_channel), nn.ReLU(inplace=True))
        self.upsample_for_pedestrian = upsample_for_pedestrian
        if upsample_for_pedestrian:
            self.upsample_conv = nn.Sequential(
                nn.ConvTranspose2d(share_conv_channel, share_conv_channel, 2, stride=2, bias=False),
                nn.BatchNorm2d(share_conv_channel), nn.ReLU())
        self.common_heads = common_heads
        self.init_bias = init_bias
        task_heads = []
        for num_cls in self.num_classes:
            heads = copy.deepcopy(dict(common_heads))
            heads.update(dict(hm=(num_cls, 2)))
            task_heads.append(heads)
        if packed_heads:
            self.tasks = PackedSepHeads(share_conv_channel, task_heads, init_bias=init_bias, final_kernel=3)
        else:
            self.tasks = nn.ModuleList([SepHead(share_conv_channel, h, bn=True, init_bias=init_bias,
                                                final_kernel=3) for h in task_heads])
        self.target_assigner = target_assigner
        self.proposal_layer = proposal_layer

    def assign_targets(self, gt_boxes):
        return self.target_assigner.assign_targets(gt_boxes)

    def forward(self, spatial_features_2d, gt_boxes=None, targets=None):
        x = self.shared_conv(spatial_features_2d)
        if self.upsample_for_pedestrian:
            x = self.upsample_conv(x)
        if isinstance(self.tasks, PackedSepHeads):
            ret = {"multi_head_features": self.tasks(x)}
        else:
            ret = {"multi_head_features": [task(x) for task in self.tasks]}
        if self.training or self.distill:
            # The reference also assigns targets for the frozen distillation teacher
            # (center_head.py:137); they are never read, so the teacher skips the work here.
            if targets is not None:
                ret.update(targets)
            elif gt_boxes is not None and not self.distill:
                ret.update(self.assign_targets(gt_boxes))
            return ret
        if self.proposal_layer is None:
            return ret
        return self.proposal_layer.generate_predicted_boxes(ret, {})

    @staticmethod
    def _sigmoid(x):
        return torch.clamp(x.sigmoid(), min=1e-4, max=1 - 1e-4)


class CenterHeadIouAware(CenterHead):
    def __init__(self, dataset_name, tasks, target_assigner, proposal_layer, out_size_factor,
                 input_channels, grid_size, point_cloud_range, code_weights, loc_weight, iou_weight,
                 share_conv_channel, common_heads, upsample_for_pedestrian=False,
                 predict_boxes_when_training=False, mode="3d", init_bias=-2.19,
                 focal_alpha=0.25, focal_gamma=2, voxel_size_xy=None, packed_heads=True):
        super().__init__(dataset_name, tasks, target_assigner, proposal_layer, input_channels,
                         grid_size, point_cloud_range, code_weights, loc_weight, share_conv_channel,
                         common_heads, upsample_for_pedestrian, predict_boxes_when_training, mode,
                         init_bias, packed_heads=packed_heads)
        self.auto_loss = AutomaticWeightedLoss(num=len(code_weights) + 2)
        self.iou_weight = iou_weight
        self.out_size_factor = out_size_factor
        # DetHead._build_losses (centerhead_fusion_exp.py:106-117)
        self.crit = FocalLoss(focal_alpha, focal_gamma)
        self.crit_reg = CenterNetRegLoss()
        self.crit_iou_aware = CenterNetRegLoss()
        self._voxel_size_xy = voxel_size_xy

    def _code_weights(self, like):
        cw = getattr(self, "_cw_cache", None)
        if cw is None or cw.device != like.device or cw.dtype != like.dtype:
            cw = self._cw_cache = like.new_tensor(self.code_weights)   # built once (no per-step H2D)
        return cw

    def _voxel_xy(self):
        if self._voxel_size_xy is not None:
            return self._voxel_size_xy
        return self.proposal_layer.voxel_size

    @staticmethod
    def local_normalisers(targets):
        """[focal num_pos per task] + [#objects per task] as 0-dim tensors (before the all-reduce)."""
        st = targets.get("_stacked")
        if st is not None:          # two reductions instead of 2T
            return list(torch.cat([st["heatmap"].eq(1).float().sum(dim=(1, 2, 3, 4)),
                                   st["mask"].float().sum(dim=(1, 2))]).unbind(0))
        T = len(targets["mask"])
        return [targets["heatmap"][t].eq(1).float().sum() for t in range(T)] + \
               [targets["mask"][t].float().sum() for t in range(T)]

    def get_loss(self, forward_ret_dict, norm=None):
        """-> (loss, tb_dict of DEVICE scalars).  Replaces each task's ``hm`` by its clamped sigmoid
        like the reference does (center_head_iou_aware.py:61) -- the response distillation relies
        on it.  ``norm``: the 2T globally averaged normalisers if the caller already reduced them.
        All tasks are evaluated together on [T, B, ...] tensors (class maps padded to the widest
        task with "ignore" labels), so the op count is independent of the number of tasks."""
        preds = forward_ret_dict["multi_head_features"]
        T = len(preds)
        st = forward_ret_dict.get("_stacked")
        if st is None:
            ncm = max(self.num_classes)
            st = {"heatmap": torch.stack([torch.nn.functional.pad(forward_ret_dict["heatmap"][t],
                                                                  (0, 0, 0, 0, 0, ncm - self.num_classes[t]))
                                          for t in range(T)]),
                  "ind": torch.stack([forward_ret_dict["ind"][t] for t in range(T)]),
                  "mask": torch.stack([forward_ret_dict["mask"][t] for t in range(T)]),
                  "box_encoding": torch.stack([forward_ret_dict["box_encoding"][t] for t in range(T)])}
        if norm is None:
            # every normaliser of the step in ONE collective: focal num_pos and #objects per task
            norm = reduce_mean_many(self.local_normalisers(forward_ret_dict))
        norm = torch.stack(list(norm)) if isinstance(norm, (list, tuple)) else norm
        num_pos_f, num_obj = norm[:T], norm[T:2 * T]
        gt_hm, ind, mask, tgt_all = st["heatmap"], st["ind"], st["mask"], st["box_encoding"]
        ncm = gt_hm.shape[2]
        B, K = ind.shape[1], ind.shape[2]
        if self.dataset == "nuscenes":
            heads, nb = ("reg", "height", "dim", "rot", "vel", "iou"), 10
        else:
            heads, nb = ("reg", "height", "dim", "rot", "iou"), 8
        a, gm = self.crit.alpha, self.crit.gamma
        stride, vs = self.out_size_factor, self._voxel_xy()
        if self.fused_loss and hiploss.supported(preds, nb):
            # ---- HIP path: focal term over all heat maps + gathered regression / IoU terms, 2 kernels
            prob, pos_l, neg_l = hiploss.focal_terms([pd["hm"] for pd in preds], gt_hm, a, gm)
            for t, pd in enumerate(preds):
                pd["hm"] = prob[t, :, :self.num_classes[t]]
            box_loss, iou_loss, iou_aware = hiploss.reg_terms(preds, ind, mask, tgt_all, num_obj,
                                                              stride * vs[0], stride * vs[1], nb)
        else:
            pos_l, neg_l, box_loss, iou_loss, iou_aware = self._loss_terms_torch(
                preds, gt_hm, ind, mask, tgt_all, num_obj, ncm, heads, nb, a, gm, stride, vs, forward_ret_dict)
        safe = torch.where(num_pos_f == 0, torch.ones_like(num_pos_f), num_pos_f)
        hm_loss = torch.where(num_pos_f == 0, -neg_l, -(pos_l + neg_l) / safe)          # [T]
        loc_loss = (box_loss * self._code_weights(box_loss)[None, :]).sum(1)                         # [T]
        p2 = self.auto_loss.params[:3] ** 2
        loss = (0.5 / p2[0] * hm_loss + 0.5 / p2[1] * loc_loss + 0.5 / p2[2] * iou_aware) \
            + torch.log(1 + p2).sum()
        # reference: ``if loc_loss.item() < 1: loss += iou_loss * w``  -> device-side select
        loss = loss + torch.where(loc_loss.detach() < 1, iou_loss * self.iou_weight, torch.zeros_like(iou_loss))
        npos = mask.float().sum(dim=(1, 2))
        tb = {}
        for t in range(T):
            key = f"task_{t}/"
            tb.update({key + "loss": loss[t].detach(), key + "hm_loss": hm_loss[t].detach(),
                       key + "loc_loss": loc_loss[t].detach(), key + "box_loss": box_loss[t].detach(),
                       key + "num_positive": npos[t]})
        return loss.sum(), tb

    fused_loss = True      # HIP detection-loss kernels for fp32 CUDA heads (ops/det_loss.py)

    def _loss_terms_torch(self, preds, gt_hm, ind, mask, tgt_all, num_obj, ncm, heads, nb, a, gm, stride, vs,
                          forward_ret_dict):
        """Tensor-op formulation (CPU / fallback / parity reference of the HIP kernels)."""
        T = len(preds)
        B, K = ind.shape[1], ind.shape[2]
        # ---- heat maps: pad every task to ncm classes; padded channels are "ignore" (gt = -1)
        logits = torch.stack([torch.nn.functional.pad(pd["hm"], (0, 0, 0, 0, 0, ncm - pd["hm"].shape[1]))
                              for pd in preds])                                      # [T,B,ncm,H,W]
        prob = self._sigmoid(logits)
        for t, pd in enumerate(preds):
            pd["hm"] = prob[t, :, :self.num_classes[t]]
        cls_valid = (torch.arange(ncm, device=prob.device)[None, :]
                     < torch.tensor(self.num_classes, device=prob.device)[:, None]) \
            if not hasattr(self, "_cls_valid") or self._cls_valid.device != prob.device else self._cls_valid
        self._cls_valid = cls_valid
        cv = cls_valid[:, None, :, None, None]
        pos = gt_hm.eq(1) & cv
        neg = gt_hm.eq(0) & cv
        pos_l = (torch.log(prob) * torch.pow(1 - prob, gm) * pos.long() * a).sum(dim=(1, 2, 3, 4))
        neg_l = (torch.log(1 - prob + 1e-4) * torch.pow(prob, gm) * neg.long() * (1 - a)).sum(dim=(1, 2, 3, 4))
        # ---- regression / IoU terms on the gathered predictions
        enc = torch.cat([torch.stack([pd[hn] for pd in preds]) for hn in heads], 2)    # [T,B,nb+1,H,W]
        forward_ret_dict["pred_box_encoding"] = {t: enc[t] for t in range(T)}
        g = _transpose_and_gather_feat(enc.reshape(T * B, nb + 1, *enc.shape[3:]), ind.reshape(T * B, K))
        g = g.reshape(T, B, K, nb + 1)
        tgt = tgt_all[..., :nb]
        iou_loss, iou_aware = self._iou_losses(g, tgt, mask, num_obj, stride, vs)        # [T], [T]
        m = mask.unsqueeze(3).float() * (~torch.isnan(tgt)).float()
        box_loss = torch.abs(g[..., :nb] * m - tgt * m).sum(dim=(1, 2)) / (num_obj[:, None] + 1e-4)   # [T,nb]
        return pos_l, neg_l, box_loss, iou_loss, iou_aware

    def _iou_losses(self, pred, tgt, mask, num_pos, stride, vs):
        """pred [T,B,K,nb+1] gathered predictions (last = iou head), tgt [T,B,K,nb], mask [T,B,K],
        num_pos [T]  ->  (iou_loss [T], iou_aware_loss [T])."""
        def decode(e):
            x = e[..., 0:1] * stride * vs[0]
            y = e[..., 1:2] * stride * vs[1]
            # clamp(exp(x), .001, 30) as in the reference; the inner clamp only keeps exp() finite so
            # that an overflowing logit gets the zero gradient of the outer clamp instead of 0*inf = NaN
            whl = torch.clamp(torch.exp(torch.clamp(e[..., 3:6], max=80.0)), min=0.001, max=30)
            rot = torch.atan2(e[..., 6:7], e[..., 7:8])
            z = e[..., 2:3]
            return x, y, z, whl, rot
        tx, ty, tz, twhl, trot = decode(tgt)
        px, py, pz, pwhl, prot = decode(pred)
        # axis-aligned "3-D IoU" with the reference's (x<->whl0, y<->whl2, z<->whl1) pairing
        def overlap(pc, pe, tc, te):
            return torch.clamp(torch.min(pc + pe / 2, tc + te / 2) - torch.max(pc - pe / 2, tc - te / 2), min=1e-3)
        ix = overlap(px, pwhl[..., 0:1], tx, twhl[..., 0:1])
        iy = overlap(py, pwhl[..., 2:3], ty, twhl[..., 2:3])
        iz = overlap(pz, pwhl[..., 1:2], tz, twhl[..., 1:2])
        inter = ix * iy * iz
        vp = torch.clamp(pwhl[..., 0:1] * pwhl[..., 2:3] * pwhl[..., 1:2], min=1e-3)
        vt = torch.clamp(twhl[..., 0:1] * twhl[..., 2:3] * twhl[..., 1:2], min=1e-3)
        iou = inter / (vp + vt - inter)
        mf = mask.unsqueeze(3).float()
        iou_loss = ((1 - torch.clamp(iou, 0, 1)) * mf).sum(dim=(1, 2, 3)) / torch.clamp_min(num_pos, 1)
        # IoU-aware target: nearest-BEV IoU between target and (detached) predicted box
        tb3 = torch.cat([tx, ty, tz, twhl, trot], -1)
        pb3 = torch.cat([px, py, pz, pwhl, prot], -1).detach()
        tar = 2 * (nearest_bev_iou_pairwise(tb3.reshape(-1, 7), pb3.reshape(-1, 7)).reshape(*mask.shape, 1) - 0.5)
        m = mf * (~torch.isnan(tar)).float()
        aware = torch.abs(pred[..., -1:] * m - tar * m).sum(dim=(1, 2, 3)) / (num_pos + 1e-4)
        return iou_loss, aware
