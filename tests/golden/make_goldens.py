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
          geom=geom, geom_xyz=geom_xyz)


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

if __name__ == "__main__":
    names = sys.argv[1:] or list(ALL)
    for n in names:
        ALL[n]()
