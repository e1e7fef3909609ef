"""Generate the golden vectors under tests/golden/*.npz from the REFERENCE python itself.

Run in the build container only (needs /root/reference):
    python tests/golden/make_goldens.py [name ...]
Each fixture holds inputs and the outputs the reference's own code produced for them on CPU
(torch 2.10, fp32), plus the seeds/shapes used.  The reference source never ships: only these
data files do.  See SURVEY.md 8c for the list; fixture <-> reference mapping is in each function.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _ref_import  # noqa: E402

_ref_import.install()

from unidistill.layers.blocks_3d.mmdet3d import lss_fpn as ref_lss  # noqa: E402
from unidistill.layers.blocks_3d.det3d.vfe.mean_vfe import MeanVFE as RefMeanVFE  # noqa: E402


def _save(name, **arrays):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **{k: (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v))
                                 for k, v in arrays.items()})
    print(f"wrote {path} ({os.path.getsize(path) / 1024:.1f} KiB)")


def _rand_calib(g, B, ncam, final_dim):
    """Random but realistic camera rigs: intrinsics, cam->ego, image-aug (ida) and bev-aug (bda)."""
    H, W = final_dim
    s2e = torch.zeros(B, ncam, 4, 4)
    intr = torch.zeros(B, ncam, 4, 4)
    ida = torch.zeros(B, ncam, 4, 4)
    base = torch.tensor([[0., 0., 1.], [-1., 0., 0.], [0., -1., 0.]])
    for b in range(B):
        for c in range(ncam):
            yaw = float(torch.rand((), generator=g)) * 6.283
            cy, sy = np.cos(yaw), np.sin(yaw)
            rz = torch.tensor([[cy, -sy, 0.], [sy, cy, 0.], [0., 0., 1.]], dtype=torch.float32)
            m = torch.eye(4)
            m[:3, :3] = rz @ base
            m[:3, 3] = torch.tensor([1.5 * cy, 0.45 * sy, 1.5]) + 0.1 * torch.randn(3, generator=g)
            s2e[b, c] = m
            k = torch.eye(4)
            f = 1266.4 * (0.9 + 0.2 * float(torch.rand((), generator=g)))
            k[0, 0] = k[1, 1] = f
            k[0, 2], k[1, 2] = 816.3, 491.5
            intr[b, c] = k
            a = torch.eye(4)
            sc = W / 1600.0 * (0.95 + 0.1 * float(torch.rand((), generator=g)))
            rot = (float(torch.rand((), generator=g)) - 0.5) * 0.1
            a[0, 0], a[0, 1] = sc * np.cos(rot), -sc * np.sin(rot)
            a[1, 0], a[1, 1] = sc * np.sin(rot), sc * np.cos(rot)
            a[0, 3], a[1, 3] = -5.0 * float(torch.rand((), generator=g)), -(900 * sc - H)
            ida[b, c] = a
    bda = torch.eye(4).repeat(B, 1, 1)
    for b in range(B):
        th = (float(torch.rand((), generator=g)) - 0.5) * 0.6
        s = 0.95 + 0.1 * float(torch.rand((), generator=g))
        bda[b, :2, :2] = s * torch.tensor([[np.cos(th), -np.sin(th)], [np.sin(th), np.cos(th)]],
                                          dtype=torch.float32)
        bda[b, 2, 2] = s
        if b % 2 == 1:
            bda[b, 0] *= -1  # flip-x
    return s2e, intr, ida, bda


def _make_lss(final_dim, x_bound, y_bound, z_bound, d_bound, out_ch, in_ch):
    return ref_lss.LSSFPN(x_bound=x_bound, y_bound=y_bound, z_bound=z_bound, d_bound=d_bound,
                          final_dim=final_dim, downsample_factor=16, output_channels=out_ch,
                          img_backbone_conf={}, img_neck_conf={},
                          depth_net_conf=dict(in_channels=in_ch))


def gold_lss_geometry():
    """create_frustum / get_geometry / binning -- lss_fpn.py:173-240, :311-313."""
    g = torch.Generator().manual_seed(101)
    final_dim = (64, 176)
    B, ncam = 2, 2
    lss = _make_lss(final_dim, [-54.0, 54.0, 0.6], [-54.0, 54.0, 0.6], [-5.0, 3.0, 8.0],
                    [2.0, 58.0, 0.5], 8, 16)
    s2e, intr, ida, bda = _rand_calib(g, B, ncam, final_dim)
    with torch.no_grad():
        geom = lss.get_geometry(s2e, intr, ida, bda)
        geom_xyz = ((geom - (lss.voxel_coord - lss.voxel_size / 2.0)) / lss.voxel_size).int()
    _save("lss_geometry", final_dim=final_dim, d_bound=[2.0, 58.0, 0.5],
          x_bound=[-54.0, 54.0, 0.6], y_bound=[-54.0, 54.0, 0.6], z_bound=[-5.0, 3.0, 8.0],
          sensor2ego=s2e, intrin=intr, ida=ida, bda=bda, frustum=lss.frustum,
          voxel_size=lss.voxel_size, voxel_coord=lss.voxel_coord, voxel_num=lss.voxel_num,
          geom=geom, geom_xyz=geom_xyz,
          # the two inverses exactly as get_geometry's own calls produce them (lss_fpn.py:222,233);
          # their last-bit rounding belongs to the LAPACK backend (MKL here), not to the algorithm
          ida_inv=ida.inverse(), intrin_inv=torch.inverse(intr))


def gold_lss_lift():
    """depth softmax (x) context -> permuted [B,ncam,D,fH,fW,C] -- lss_fpn.py:289-310."""
    g = torch.Generator().manual_seed(102)
    B, ncam, D, C, fH, fW = 1, 2, 16, 8, 4, 11
    depth_feature = torch.randn(B * ncam, D + C, fH, fW, generator=g)
    depth = depth_feature[:, :D].softmax(1)
    feat = depth.unsqueeze(1) * depth_feature[:, D:D + C].unsqueeze(2)
    feat = feat.reshape(B, ncam, C, D, fH, fW).permute(0, 1, 3, 4, 5, 2).contiguous()
    _save("lss_lift", depth_feature=depth_feature, D=D, C=C, B=B, ncam=ncam, lifted=feat,
          depth=depth)


def gold_bev_pool():
    """VoxelPooling.backward (reference python, lss_fpn.py:64-79) on random pos_memo, and the
    forward call-site contract (lss_fpn.py:43-62) through the index_add stand-in."""
    g = torch.Generator().manual_seed(103)
    B, N, C, nx, ny, nz = 2, 700, 12, 9, 7, 1
    geom = torch.stack([torch.randint(-2, nx + 2, (B, N), generator=g),
                        torch.randint(-2, ny + 2, (B, N), generator=g),
                        torch.randint(-1, nz + 1, (B, N), generator=g)], -1).int()
    feat = torch.randn(B, N, C, generator=g, requires_grad=True)
    voxel_num = torch.tensor([nx, ny, nz])
    out = ref_lss.voxel_pooling(geom, feat, voxel_num)       # [B, C, ny, nx] view
    gout = torch.randn(out.shape, generator=g)
    out.backward(gout)
    # recover pos_memo the way the Function stored it
    pos = torch.full((B, N, 3), -1, dtype=torch.int32)
    kept = (geom[..., 0] >= 0) & (geom[..., 0] < nx) & (geom[..., 1] >= 0) & (geom[..., 1] < ny) \
        & (geom[..., 2] >= 0) & (geom[..., 2] < nz)
    bidx = torch.arange(B).view(B, 1).expand(B, N)
    pos[kept] = torch.stack([bidx[kept].int(), geom[..., 1][kept], geom[..., 0][kept]], -1)
    _save("bev_pool", geom=geom, feat=feat, nx=nx, ny=ny, nz=nz, out_nchw=out, gout_nchw=gout,
          gfeat=feat.grad, pos=pos)


def gold_mean_vfe():
    """MeanVFE.forward -- layers/blocks_3d/det3d/vfe/mean_vfe.py:14-34 (incl. zero-count voxels)."""
    g = torch.Generator().manual_seed(108)
    M, P, F = 257, 10, 5
    num = torch.randint(0, P + 1, (M,), generator=g).int()
    vox = torch.randn(M, P, F, generator=g)
    mask = torch.arange(P).view(1, P) < num.view(M, 1)
    vox = vox * mask.unsqueeze(-1)
    out = RefMeanVFE(F)(vox, num)
    _save("mean_vfe", voxels=vox, num=num, out=out)


ALL = {
    "lss_geometry": gold_lss_geometry,
    "lss_lift": gold_lss_lift,
    "bev_pool": gold_bev_pool,
    "mean_vfe": gold_mean_vfe,
}



def gold_distill():
    """FeatureDistillLoss / BEVDistillLoss / ResponseDistillLoss (+ gaussian mask, box corners)
    forward values and student-side grads -- BEVFusion_nuscenes_centerhead_camera_exp_distill_lidar.py
    :73-97 (center_to_corner_box2d), :100-178 (gaussian mask), :196-245, :248-323, :326-385,
    and the coordinate prep of training_step :466-483.  1e-4 and 1e-3 sigmoid clamps (:191-193)."""
    import importlib
    base = "unidistill.exps.multisensor_fusion.nuscenes.BEVFusion."
    mod_a = importlib.import_module(base + "BEVFusion_nuscenes_centerhead_camera_exp_distill_lidar")
    mod_b = importlib.import_module(base + "BEVFusion_nuscenes_centerhead_camera_exp_distill_fusion")
    g = torch.Generator().manual_seed(107)
    B, M, H, W = 2, 6, 20, 20
    pc_range = [-6.0, -6.0, -5.0, 6.0, 6.0, 3.0]
    voxel = [0.075, 0.075, 0.2]
    osf = 8                                  # 0.6 m per BEV pixel -> 20 px over 12 m
    gt = torch.zeros(B, M, 10)
    nvalid = [5, 3]
    for b in range(B):
        n = nvalid[b]
        gt[b, :n, 0:2] = (torch.rand(n, 2, generator=g) - 0.5) * 11.0
        gt[b, :n, 2] = torch.randn(n, generator=g) * 0.5 - 1.0
        gt[b, :n, 3:6] = torch.rand(n, 3, generator=g) * 3.0 + 0.5
        gt[b, :n, 6] = (torch.rand(n, generator=g) - 0.5) * 6.28
        gt[b, :n, 7:9] = torch.randn(n, 2, generator=g)
        gt[b, :n, 9] = torch.randint(1, 11, (n,), generator=g).float()
    gt[0, 0, 0:2] = torch.tensor([5.7, -5.8])        # a box hanging over the map border
    # training_step's valid-box scan (:449-455): trailing zero rows are invalid
    idx = torch.zeros(B, M)
    for i in range(B):
        cnt = M - 1
        while cnt > 0 and gt[i][cnt].sum() == 0:
            cnt -= 1
        idx[i][:cnt + 1] = 1
    idx = idx.bool()
    corners = torch.zeros(B, M, 4, 2)
    for i in range(B):
        corners[i] = mod_a.center_to_corner_box2d(gt[i][:, :2].numpy(), gt[i][:, 3:5].numpy(),
                                                  gt[i][:, 6].numpy(), origin=(0.5, 0.5))
    corners_m = corners.clone()
    corners[..., 0] = (corners[..., 0] - pc_range[0]) / (voxel[0] * osf)
    corners[..., 1] = (corners[..., 1] - pc_range[1]) / (voxel[1] * osf)
    out = dict(gt=gt, valid=idx, corners_m=corners_m, corners_px=corners, pc_range=pc_range,
               voxel=voxel, osf=osf)
    # feature / relation losses
    for name, fn, C in (("feat", mod_a.FeatureDistillLoss, 16), ("rel", mod_a.BEVDistillLoss, 24)):
        s = torch.randn(B, C, H, W, generator=g, requires_grad=True)
        t = torch.randn(B, C, H, W, generator=g)
        loss = fn(s, t, corners.clone(), idx)
        loss.backward()
        out.update({f"{name}_s": s.detach(), f"{name}_t": t, f"{name}_loss": loss.detach(),
                    f"{name}_grad": s.grad})
    # response loss: 2 tasks with 1 and 2 classes
    heads = (("reg", 2), ("height", 1), ("dim", 3), ("rot", 2), ("vel", 2), ("iou", 1))
    ncls = (1, 2)
    for tag, mod in (("a", mod_a), ("b", mod_b)):
        rs, rt, leaves = [], [], []
        for ti, nc in enumerate(ncls):
            ds, dt = {}, {}
            hm_logit = torch.randn(B, nc, H, W, generator=g, requires_grad=True)
            leaves.append(hm_logit)
            ds["hm"] = mod._sigmoid(hm_logit)            # student hm is post-sigmoid (quirk 1)
            dt["hm"] = torch.randn(B, nc, H, W, generator=g) * 2.0
            for hn, hc in heads:
                v = torch.randn(B, hc, H, W, generator=g, requires_grad=True)
                leaves.append(v)
                ds[hn] = v
                dt[hn] = torch.randn(B, hc, H, W, generator=g)
            rs.append(ds)
            rt.append(dt)
        lc, lr = mod.ResponseDistillLoss(rs, rt, gt, pc_range, voxel, osf)
        (lc + 2.0 * lr).backward()
        mask = mod.calculate_box_mask_gaussian((B, 22, H, W), gt.numpy(), pc_range, voxel, osf)
        k = 0
        for ti in range(len(ncls)):
            for hn in ("hm",) + tuple(h for h, _ in heads):
                out[f"resp_{tag}_s{ti}_{hn}"] = (rs[ti][hn] if hn != "hm" else leaves[k]).detach()
                out[f"resp_{tag}_t{ti}_{hn}"] = rt[ti][hn]
                out[f"resp_{tag}_g{ti}_{hn}"] = leaves[k].grad
                k += 1
        out.update({f"resp_{tag}_cls": lc.detach(), f"resp_{tag}_reg": lr.detach(), f"resp_{tag}_mask": mask})
    _save("distill", **out)


ALL["distill"] = gold_distill


def _ref_head(tasks, share, in_ch, grid, pc_range, voxel, osf, max_objs=200):
    """Build the reference CenterHeadIouAware + FCOSAssigner exactly like DetHead.build_dense_head
    (BEVFusion_nuscenes_centerhead_fusion_exp.py:51-119) for a shrunk configuration."""
    from unidistill.layers.head.det3d import CenterHeadIouAware, FCOSAssigner
    from unidistill.layers.losses.det3d import CenterNetRegLoss, FocalLoss
    names = [n for t in tasks for n in t["class_names"]]
    tasks_cfg = [_ref_import._AttrDict(t) for t in tasks]
    assigner = FCOSAssigner(out_size_factor=osf, tasks=tasks_cfg, dense_reg=1, gaussian_overlap=0.1,
                            max_objs=max_objs, min_radius=2,
                            mapping={n: i + 1 for i, n in enumerate(names)}, grid_size=grid,
                            pc_range=pc_range[0:2], voxel_size=voxel[0:2], assign_topk=9,
                            no_log=False, with_velocity=True)

    class _Prop:                      # only .voxel_size / .training are touched in training
        voxel_size = voxel[0:2]
        training = True
    head = CenterHeadIouAware(
        dataset_name="nuscenes", tasks=tasks_cfg, target_assigner=assigner, proposal_layer=_Prop(),
        out_size_factor=osf, input_channels=in_ch, grid_size=grid, point_cloud_range=pc_range,
        code_weights=[1.0] * 8 + [0.2, 0.2], loc_weight=0.25, iou_weight=5.0,
        share_conv_channel=share,
        common_heads={"iou": [1, 2], "reg": [2, 2], "height": [1, 2], "dim": [3, 2], "rot": [2, 2],
                      "vel": [2, 2]},
        upsample_for_pedestrian=False, mode="3d", init_bias=-2.19, predict_boxes_when_training=False)
    head.add_module("crit", FocalLoss(0.25, 2))
    head.add_module("crit_reg", CenterNetRegLoss())
    head.add_module("crit_iou_aware", CenterNetRegLoss())
    return head


def _rand_gt(g, B, M, nvalid, ncls, lo, hi):
    gt = torch.zeros(B, M, 10)
    for b in range(B):
        n = nvalid[b]
        gt[b, :n, 0:2] = lo + (hi - lo) * torch.rand(n, 2, generator=g)
        gt[b, :n, 2] = torch.randn(n, generator=g) * 0.5 - 1.0
        gt[b, :n, 3:6] = torch.rand(n, 3, generator=g) * 3.0 + 0.4
        gt[b, :n, 6] = (torch.rand(n, generator=g) - 0.5) * 12.0
        gt[b, :n, 7:9] = torch.randn(n, 2, generator=g)
        gt[b, :n, 9] = torch.randint(1, ncls + 1, (n,), generator=g).float()
    return gt


def gold_dense_head():
    """BaseBEVBackbone (base_bev_backbone.py:10-174), CenterHeadIouAware forward + FCOSAssigner
    targets + get_loss (center_head.py:124-146, fcos_assigner.py:73-285,
    center_head_iou_aware.py:55-298), FocalLoss / CenterNetRegLoss (losses/det3d.py:287-421),
    boxes3d_nearest_bev_iou (box_utils.py:343-373) -- shrunk widths, seeded state_dicts."""
    from unidistill.layers.blocks_2d.det3d import BaseBEVBackbone
    from unidistill.utils.det3d_utils import box_utils
    torch.manual_seed(104)
    g = torch.Generator().manual_seed(104)
    out = {}
    # ---- trunk
    trunk = BaseBEVBackbone(layer_nums=[2, 2], layer_strides=[1, 2], num_filters=[8, 16],
                            upsample_strides=[1, 2], num_upsample_filters=[12, 12], input_channels=6)
    for m in trunk.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.normal_(0, 0.2)
            m.running_var.uniform_(0.5, 1.5)
            m.weight.data.uniform_(0.5, 1.5)
            m.bias.data.normal_(0, 0.2)
    x = torch.randn(2, 6, 32, 32, generator=g)
    for k, v in trunk.state_dict().items():         # before the train-mode pass moves the BN stats
        out["trunk_sd/" + k] = v.clone()
    trunk.eval()
    with torch.no_grad():
        y_eval, pyr = trunk(x)
    trunk.train()
    y_train, _ = trunk(x)
    out.update({"trunk_x": x, "trunk_y_eval": y_eval, "trunk_y_train": y_train.detach(),
                "trunk_pyr2": pyr["spatial_features_2x"]})
    # ---- head + assigner + loss  (32x32 map, 8 voxels per pixel, 0.25 m voxels -> 64 m square)
    tasks = [dict(num_class=1, class_names=["car"]), dict(num_class=2, class_names=["truck", "bus"]),
             dict(num_class=1, class_names=["barrier"])]
    pc_range = [-32.0, -32.0, -5.0, 32.0, 32.0, 3.0]
    voxel = [0.25, 0.25, 0.2]
    head = _ref_head(tasks, share=16, in_ch=24, grid=[256, 256, 40], pc_range=pc_range, voxel=voxel,
                     osf=8)
    for m in head.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.normal_(0, 0.2)
            m.running_var.uniform_(0.5, 1.5)
    with torch.no_grad():
        head.auto_loss.params.copy_(torch.linspace(0.8, 1.3, 12))
    B, M = 2, 12
    gt = _rand_gt(g, B, M, [9, 0], 3, -30.0, 30.0)      # sample 1 has no boxes at all; class 4 absent
    gt[0, 3, 3:6] = 0.0                                    # a zero-size box -> log(0) = -inf -> zeroed
    feat = torch.randn(B, 24, 32, 32, generator=g, requires_grad=True)
    for k, v in head.state_dict().items():
        out["head_sd/" + k] = v.clone()
    head.train()
    ret = head(feat, gt.clone())
    for _, enc in ret["box_encoding"].items():
        enc[torch.isinf(enc)] = 0
    loss, tb = head.get_loss(ret)
    loss.backward()
    out.update({"head_feat": feat.detach(), "head_gt": gt, "head_loss": loss.detach(),
                "head_feat_grad": feat.grad, "head_params_grad": head.auto_loss.params.grad})
    for t in range(len(tasks)):
        for k in ("heatmap", "ind", "mask", "cat", "box_encoding"):
            out[f"head_tgt{t}_{k}"] = ret[k][t]
        for hn, v in ret["multi_head_features"][t].items():
            out[f"head_out{t}_{hn}"] = v.detach()      # hm is post-sigmoid after get_loss (quirk 1)
        out[f"head_tb{t}"] = np.array([tb[f"task_{t}/{k}"] for k in
                                       ("loss", "hm_loss", "loc_loss", "x_loss", "vy_loss")], np.float64)
    # ---- standalone box helpers
    a = torch.randn(40, 7, generator=g)
    a[:, 3:6] = a[:, 3:6].abs() + 0.3
    b = a + 0.3 * torch.randn(40, 7, generator=g)
    b[:, 3:6] = b[:, 3:6].abs() + 0.3
    out.update({"iou_a": a, "iou_b": b, "iou_bev": box_utils.boxes3d_nearest_bev_iou(a, b)})
    _save("dense_head", **out)


ALL["dense_head"] = gold_dense_head


def gold_proposals():
    """IouAwareGenProposals.generate_predicted_boxes (iou_aware_gen_proposals.py:43-139,
    centerpoint_gen_proposals.py:85-340): top-K decode, range/score filter, NMS, roi padding.  The
    reference's NMS binary is missing from its tree; the harness binds `iou3d_nms_cuda.nms_gpu` to the
    CPU oracle (oracle.nms_bev), so this golden pins everything AROUND the NMS against the reference's own
    code and the NMS itself against the oracle."""
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    import oracle
    from unidistill.layers.head.det3d.generate_proposals import iou3d_nms_cuda, IouAwareGenProposals

    def nms_gpu(boxes, keep, thresh):
        kept = oracle.nms_bev(boxes.detach().cpu().numpy(), float(thresh))
        keep[:len(kept)] = torch.from_numpy(kept)
        return len(kept)
    iou3d_nms_cuda.nms_gpu = nms_gpu
    g = torch.Generator().manual_seed(105)
    tasks = [["car"], ["truck", "bus"], ["barrier"]]
    prop = IouAwareGenProposals(
        dataset_name="nuscenes", class_names=tasks, post_center_limit_range=[-30.0, -30.0, -6.0, 30.0, 30.0, 6.0],
        score_threshold=0.1, pc_range=[-32.0, -32.0], out_size_factor=8, voxel_size=[0.25, 0.25], no_log=False,
        iou_aware_list=[0.65] * 3, nms_iou_threshold_train=0.8, nms_pre_max_size_train=60,
        nms_post_max_size_train=20, nms_iou_threshold_test=0.2, nms_pre_max_size_test=50,
        nms_post_max_size_test=12)
    B, H, W = 2, 32, 32
    out, heads = {}, []
    for t, names in enumerate(tasks):
        d = {"hm": torch.randn(B, len(names), H, W, generator=g) * 2.0 - 1.0,
             "reg": torch.rand(B, 2, H, W, generator=g), "height": torch.randn(B, 1, H, W, generator=g),
             "dim": torch.randn(B, 3, H, W, generator=g) * 0.4 + 0.8, "rot": torch.randn(B, 2, H, W, generator=g),
             "vel": torch.randn(B, 2, H, W, generator=g), "iou": torch.randn(B, 1, H, W, generator=g)}
        heads.append(d)
        for k, v in d.items():
            out[f"in{t}_{k}"] = v
    for phase in ("train", "test"):
        prop.training = phase == "train"
        res = prop.generate_predicted_boxes({"multi_head_features": [dict(d) for d in heads]}, {})
        out[f"{phase}_rois"] = res["rois"]
        out[f"{phase}_roi_scores"] = res["roi_scores"]
        out[f"{phase}_roi_labels"] = res["roi_labels"]
        for b, pd in enumerate(res["pred_dicts"]):
            out[f"{phase}_n{b}"] = np.array([pd["pred_boxes"].shape[0]])
    _save("proposals", **out)


ALL["proposals"] = gold_proposals

def gold_input_prep():
    """LiDAR input side: CollectLidarSweeps.forward (data/multisensorfusion/transforms3d.py:379-414:
    sweeps into the key frame, time-lag channel) and BevAffineTransformation.forward (:417-443) with
    functional.bev_transform (functional.py:595-646: rotate / scale / translate / flip of points and boxes).
    The random augmentation draw is pinned by overriding sample_augs; everything else is the reference's code."""
    from unidistill.data.multisensorfusion import transforms3d as T
    rng = np.random.default_rng(77)

    def pose():
        a = rng.uniform(-np.pi, np.pi)
        c, s_ = np.cos(a), np.sin(a)
        tilt = rng.normal(scale=0.02, size=2)
        rx = np.array([[1, 0, 0], [0, np.cos(tilt[0]), -np.sin(tilt[0])], [0, np.sin(tilt[0]), np.cos(tilt[0])]])
        ry = np.array([[np.cos(tilt[1]), 0, np.sin(tilt[1])], [0, 1, 0], [-np.sin(tilt[1]), 0, np.cos(tilt[1])]])
        m = np.eye(4)
        m[:3, :3] = np.array([[c, -s_, 0], [s_, c, 0], [0, 0, 1]]) @ rx @ ry
        m[:3, 3] = rng.normal(scale=[300.0, 300.0, 1.0])
        return m

    def cloud(n):
        pts = np.zeros((n, 5), dtype=np.float32)
        pts[:, :3] = rng.normal(scale=[25.0, 25.0, 1.5], size=(n, 3))
        pts[:, 3] = rng.uniform(0, 255, size=n)
        pts[:, 4] = rng.uniform(0, 31, size=n)          # ring index: overwritten by the time lag
        return pts

    out = {}
    key_l2e, key_e2g = pose(), pose()
    key_l2e[:3, 3] = [0.94, 0.0, 1.84]
    sizes = [1500, 1200, 0, 977]
    sweeps = [cloud(n) for n in sizes[1:]]
    sweep_infos, t0 = [], 1533151603547590
    for i in range(len(sweeps)):
        m = key_e2g.copy()
        m[:3, 3] += rng.normal(scale=[1.5, 1.5, 0.02])                      # ego moved a little between sweeps
        sweep_infos.append({"sweep_lidar_to_ego": m, "sweep_lidar_timestamp": t0 - 50000 * (i + 1) - int(rng.integers(0, 900))})
    key = cloud(sizes[0])
    out["key_points"], out["key_lidar_to_ego"], out["key_ego_to_global"] = key.copy(), key_l2e, key_e2g
    out["timestamp"] = np.array([t0], dtype=np.int64)
    for i, (sw, inf) in enumerate(zip(sweeps, sweep_infos)):
        out[f"sweep{i}_points"] = sw.copy()
        out[f"sweep{i}_lidar_to_ego"] = inf["sweep_lidar_to_ego"]
        out[f"sweep{i}_timestamp"] = np.array([inf["sweep_lidar_timestamp"]], dtype=np.int64)
    dd = {"points": key.copy(), "sweep_points": [s_.copy() for s_ in sweeps],
          "info": {"ego_to_global": key_e2g, "lidar_to_ego": key_l2e, "timestamp": t0,
                   "sweep_lidar_infos": [dict(d) for d in sweep_infos]}}
    dd = T.CollectLidarSweeps().forward(dd)
    out["collected_points"] = dd["points"]
    assert dd["points"].dtype == np.float32 and dd["points"].shape == (sum(sizes), 5)

    boxes = np.zeros((7, 9), dtype=np.float32)
    boxes[:, :3] = rng.normal(scale=[20.0, 20.0, 1.0], size=(7, 3))
    boxes[:, 3:6] = rng.uniform(0.5, 5.0, size=(7, 3))
    boxes[:, 6] = rng.uniform(-np.pi, np.pi, size=7)
    boxes[:, 7:] = rng.normal(scale=3.0, size=(7, 2))
    out["gt_boxes_in"] = boxes.copy()
    cases = [(11.25, 1.04, np.array([0.3, -0.2, 0.05]), False, False), (-20.0, 0.93, np.array([0.0, 0.0, 0.0]), True, False),
             (5.5, 1.0, np.array([-0.4, 0.1, 0.0]), False, True), (0.0, 1.07, np.array([0.2, 0.2, -0.1]), True, True)]
    for ci, augs in enumerate(cases):
        bda = T.BevAffineTransformation(rot_lim=(-22.5, 22.5), scale_lim=(0.9, 1.1), trans_lim=(0.5, 0.5, 0.5),
                                        flip_dx_ratio=0.5, flip_dy_ratio=0.5)
        bda.sample_augs = lambda augs=augs: augs
        d2 = {"gt_boxes": boxes.copy(), "points": dd["points"].copy(), "imgs": {}}
        d2 = bda.forward(d2)
        out[f"bda{ci}_augs"] = np.array([augs[0], augs[1], *augs[2], float(augs[3]), float(augs[4])])
        out[f"bda{ci}_points"], out[f"bda{ci}_boxes"], out[f"bda{ci}_mat"] = d2["points"], d2["gt_boxes"], d2["bda_mat"]
    _save("input_prep", **out)


ALL["input_prep"] = gold_input_prep



def gold_fusion_encoder():
    """FusionEncoder(use_elementwise=False) -- BEVFusion_nuscenes_base_exp.py:107-135: cat -> channel attention
    (global average pool -> 1x1 conv -> sigmoid) -> 3x3 conv + BN + ReLU; eval and train forward, input grads.
    128 -> 64 channels so that the MFMA conv path (channel multiples of 64) is exercised on the GPU."""
    from unidistill.exps.multisensor_fusion.nuscenes.BEVFusion.BEVFusion_nuscenes_base_exp import FusionEncoder
    torch.manual_seed(108)
    g = torch.Generator().manual_seed(108)
    m = FusionEncoder(use_elementwise=False, input_channel=128, output_channel=64)
    bn = m.reduce_conv[1]
    bn.running_mean.normal_(0, 0.2)
    bn.running_var.uniform_(0.5, 1.5)
    bn.weight.data.uniform_(0.5, 1.5)
    bn.bias.data.normal_(0, 0.2)
    out = {"sd/" + k: v.clone() for k, v in m.state_dict().items()}
    x1 = torch.randn(2, 64, 10, 12, generator=g, requires_grad=True)
    x2 = torch.randn(2, 64, 10, 12, generator=g, requires_grad=True)
    m.eval()
    with torch.no_grad():
        y_eval = m(x1, x2)
    m.train()
    y = m(x1, x2)
    gy = torch.randn(y.shape, generator=g)
    y.backward(gy)
    out.update(x1=x1.detach(), x2=x2.detach(), y_eval=y_eval, y_train=y.detach(), gy=gy, g1=x1.grad, g2=x2.grad,
               gw=m.reduce_conv[0].weight.grad, gatt=m.att[1].weight.grad,
               y_sum=FusionEncoder(use_elementwise=True)(x1.detach(), x2.detach()))
    _save("fusion_encoder", **out)


ALL["fusion_encoder"] = gold_fusion_encoder


def gold_model_step():
    """Composition golden at a shrunk size (tests/golden/shrunk.py), everything executed by the reference's own
    code on CPU:
      * LSSFPN._forward_single_sweep (lss_fpn.py:266-320) incl. the lifted [B,N,C] tensor it hands to the
        pooling extension (captured at the call site, lss_fpn.py:48-59) and the returned depth;
      * BEVFusionCenterHead.forward, training and return_feature modes (centerhead_fusion_exp.py:134-171);
      * Exp.training_step of the distillation experiment (camera_exp_distill_lidar.py:438-513): valid-box scan,
        label +1, teacher call, corner scaling, the three distillation losses and the 100 / 40 / 10 weights --
        called unbound on a stand-in ``self`` that owns the (reference) student and teacher models.
    The image backbone / neck are small stand-ins registered on both sides (mmdet is not available)."""
    import importlib
    import shrunk as S
    base = "unidistill.exps.multisensor_fusion.nuscenes.BEVFusion."
    fus = importlib.import_module(base + "BEVFusion_nuscenes_centerhead_fusion_exp")
    dis = importlib.import_module(base + "BEVFusion_nuscenes_centerhead_camera_exp_distill_lidar")
    ref_lss.build_backbone = lambda cfg: S.TinyBackbone()
    ref_lss.build_neck = lambda cfg: S.TinyNeck()
    captured = {}
    ext = sys.modules["unidistill.layers.blocks_3d.mmdet3d.voxel_pooling_ext"]

    def capture(B, N, C, nx, ny, nz, geom, feat, out_, pos):
        captured["lifted"], captured["geom_xyz"] = feat.detach().clone(), geom.detach().clone()
        return _ref_import._cpu_voxel_pooling_forward_wrapper(B, N, C, nx, ny, nz, geom, feat, out_, pos)
    ref_lss.voxel_pooling_ext.voxel_pooling_forward_wrapper = capture

    def build(seed):
        torch.manual_seed(seed)
        cfg = S.reference_model_cfg()
        for part in ("target_assigner", "proposal_layer", "dense_head"):     # mmcv.Config wraps list items too
            cfg["det_head"][part]["densehead_tasks"] = [_ref_import._AttrDict(t)
                                                        for t in cfg["det_head"][part]["densehead_tasks"]]
        m = fus.BEVFusionCenterHead(model_cfg=_ref_import._AttrDict(cfg))
        gg = torch.Generator().manual_seed(seed)
        for mod in m.modules():                      # non-trivial BN state so eval != train != identity
            if isinstance(mod, torch.nn.BatchNorm2d):
                mod.running_mean.normal_(0, 0.1, generator=gg)
                mod.running_var.uniform_(0.6, 1.4, generator=gg)
                mod.weight.data.uniform_(0.7, 1.3, generator=gg)
                mod.bias.data.normal_(0, 0.1, generator=gg)
        return m
    student, teacher = build(109), build(110)
    out = {}
    for tag, m in (("student", student), ("teacher", teacher)):
        for k, v in m.state_dict().items():
            out[f"{tag}_sd/{k}"] = v.clone()
    g = torch.Generator().manual_seed(111)
    B, ncam, M = 2, 2, 8
    imgs = torch.randn(B, 1, ncam, 3, *S.IMG_DIM, generator=g)
    s2e, intr, ida, bda = _rand_calib(g, B, ncam, S.IMG_DIM)
    mats = {"sensor2ego_mats": s2e.unsqueeze(1), "intrin_mats": intr.unsqueeze(1), "ida_mats": ida.unsqueeze(1),
            "sensor2sensor_mats": torch.eye(4).repeat(B, 1, ncam, 1, 1), "bda_mat": bda}
    gt = _rand_gt(g, B, M, [6, 3], 4, -11.0, 11.0)
    gt[0, 2] = 0.0                                   # an all-zero row before the last valid one stays "valid"
    gt_boxes, gt_labels = gt[..., :9].clone(), gt[..., 9].clone() - 1.0     # loader labels are 0-based
    out.update(imgs=imgs, sensor2ego=s2e, intrin=intr, ida=ida, bda=bda, gt_boxes=gt_boxes, gt_labels=gt_labels,
               ida_inv=ida.inverse(), intrin_inv=torch.inverse(intr))
    # ---- LSSFPN._forward_single_sweep on the student's camera encoder (eval statistics: deterministic BN)
    lss = student.camera_encoder.backbone
    student.eval()
    with torch.no_grad():
        bev, depth = lss._forward_single_sweep(0, imgs, mats, is_return_depth=True)
    out.update(lss_bev=bev, lss_depth=depth, lss_lifted=captured["lifted"], lss_geom_xyz=captured["geom_xyz"])
    # ---- the distillation training step (student in train mode, frozen teacher in eval)
    teacher.det_head.dense_head.distill = True
    for p_ in teacher.parameters():
        p_.requires_grad = False
    teacher.eval()
    student.train()
    with torch.no_grad():
        tf, tb, tr = teacher(None, imgs, mats, torch.cat([gt_boxes, (gt_labels + 1).unsqueeze(2)], 2),
                             return_feature=True)
    out.update(teacher_feat=tf, teacher_head0_hm=tr[0]["hm"])        # spot checks of the return_feature mode

    class _Self:
        """what Exp.training_step touches on ``self``"""
        def __init__(self):
            self.teacher_model, self.checkpoint_state_dict = teacher, teacher.state_dict()
            self.calls = {}

        def __call__(self, *a, **k):
            r = student(*a, **k)
            self.calls["student"] = r
            return r
    for name in ("_POINT_CLOUD_RANGE", "_VOXEL_SIZE", "_OUT_SIZE_FACTOR", "_GRID_SIZE"):
        setattr(dis, name, {"_POINT_CLOUD_RANGE": S.PCR, "_VOXEL_SIZE": S.VOXEL, "_OUT_SIZE_FACTOR": S.OSF,
                            "_GRID_SIZE": S.GRID}[name])
    me = _Self()
    batch = {"imgs": imgs, "mats_dict": mats, "gt_boxes": gt_boxes.clone(), "gt_labels": gt_labels.clone()}
    loss = dis.Exp.training_step(me, batch)
    loss.backward()
    ret, tbd, feat_s, trunk_s, heads_s, _ = me.calls["student"]
    out.update(loss=loss.detach(), loss_rpn=ret["loss"].detach(), student_feat=feat_s.detach(),
               student_trunk_mean=trunk_s.detach().mean((0, 2, 3)), student_trunk_std=trunk_s.detach().std((0, 2, 3)))
    for k in ("loss_feature", "loss_bev_rel", "loss_resp_cls", "loss_resp_reg"):
        out[k] = tbd[k].detach()
    for t in range(len(S.TASKS)):
        for hn, v in heads_s[t].items():
            out[f"student_head{t}_{hn}"] = v.detach()     # hm is post-sigmoid after get_loss (quirk 1)
    for n, p_ in student.named_parameters():
        if p_.grad is not None:
            out[f"grad/{n}"] = p_.grad.clone()
    for k, v in student.state_dict().items():             # BN running statistics after the train-mode pass
        if "running_" in k:
            out[f"student_after/{k}"] = v.clone()
    _save("model_step", **out)

    # ---- the same step with the teacher in TRAIN mode: Exp.__init__ calls teacher_model.eval()
    # (camera_exp_distill_lidar.py:424) but the teacher is a registered submodule, so Lightning's model.train() before the
    # training loop puts it back into train mode -- its BatchNorms then normalise with BATCH statistics and move their running
    # statistics, which training_step's per-step load_state_dict(self.checkpoint_state_dict) (:463) resets from the checkpoint
    # (a separate copy of the tensors, as loaded from the file).  Same weights, inputs and seeds as above.
    student2, teacher2 = build(109), build(110)
    teacher2.det_head.dense_head.distill = True
    for p_ in teacher2.parameters():
        p_.requires_grad = False
    ckpt = {k: v.clone() for k, v in teacher2.state_dict().items()}

    class _Lightning(torch.nn.Module):          # what `model.train()` reaches in the reference's LightningModule
        def __init__(self):
            super().__init__()
            self.model, self.teacher_model = student2, teacher2
    teacher2.eval()                             # Exp.__init__ (:424)
    _Lightning().train()                        # trainer.fit -> model.train()
    assert teacher2.training and student2.training

    class _Self2:
        def __init__(self):
            self.teacher_model, self.checkpoint_state_dict, self.calls = teacher2, ckpt, {}

        def __call__(self, *a, **k):
            r = student2(*a, **k)
            self.calls["student"] = r
            return r
    me2 = _Self2()
    batch2 = {"imgs": imgs, "mats_dict": mats, "gt_boxes": gt_boxes.clone(), "gt_labels": gt_labels.clone()}
    loss2 = dis.Exp.training_step(me2, batch2)
    loss2.backward()
    ret2, tbd2 = me2.calls["student"][:2]
    out2 = {"loss": loss2.detach(), "loss_rpn": ret2["loss"].detach()}
    for k in ("loss_feature", "loss_bev_rel", "loss_resp_cls", "loss_resp_reg"):
        out2[k] = tbd2[k].detach()
    assert abs(float(out2["loss_feature"]) - float(out["loss_feature"])) > 1e-3 * abs(float(out["loss_feature"])), \
        "the train-mode teacher must change the distillation terms"
    for n, p_ in student2.named_parameters():
        if p_.grad is not None:
            out2[f"grad/{n}"] = p_.grad.clone()
    for k, v in teacher2.state_dict().items():       # one momentum step away from the checkpoint (reset comes with the NEXT step)
        if "running_" in k or "num_batches_tracked" in k:
            out2[f"teacher_after/{k}"] = v.clone()
    _save("model_step_teacher_train", **out2)


ALL["model_step"] = gold_model_step


def gold_collate():
    """collate_fn (data/multisensorfusion/nuscenes_multimodal.py:418-495) executed by the reference on ragged
    samples: equal-size images / matrices are stacked, clouds and boxes zero-padded, an empty box list handled."""
    from unidistill.data.multisensorfusion.nuscenes_multimodal import collate_fn
    rng = np.random.default_rng(91)
    n_pts, n_box = [37, 52, 11], [4, 0, 7]
    data, out = [], {}
    for i, (n, m) in enumerate(zip(n_pts, n_box)):
        d = {"imgs": rng.standard_normal((1, 2, 3, 6, 10)).astype(np.float32),
             "points": rng.standard_normal((n, 5)).astype(np.float32),
             "gt_boxes": rng.standard_normal((m, 9)).astype(np.float32),
             "gt_labels": rng.integers(0, 10, (m,)).astype(np.int64),
             "mats_dict": {"sensor2ego_mats": rng.standard_normal((1, 2, 4, 4)), "intrin_mats": rng.standard_normal((1, 2, 4, 4)),
                           "ida_mats": rng.standard_normal((1, 2, 4, 4)), "sensor2sensor_mats": rng.standard_normal((1, 2, 4, 4)),
                           "bda_mat": rng.standard_normal((4, 4))},
             "img_metas": {"token": f"t{i}"}}
        data.append(d)
        for k in ("imgs", "points", "gt_boxes", "gt_labels"):
            out[f"in{i}_{k}"] = d[k]
        for k, v in d["mats_dict"].items():
            out[f"in{i}_{k}"] = v
    res = collate_fn(data)
    for k in ("imgs", "points", "gt_boxes", "gt_labels"):
        out["out_" + k] = res[k]
    for k, v in res["mats_dict"].items():
        out["out_" + k] = v
    _save("collate", **out)


ALL["collate"] = gold_collate

if __name__ == "__main__":
    names = sys.argv[1:] or list(ALL)
    for n in names:
        ALL[n]()
