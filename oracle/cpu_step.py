"""CPU restatement of BASELINE.json configs[0] -- camera-only student, 1 camera 256x704, batch 1, random
weights, forward + backward -- for bench.py's ``cpu_baseline`` leg ("torch CPU ops for
trunk / head / losses, i.e. the same math the reference runs through torch + the CPU oracle for the natives").

TEST / BASELINE INFRASTRUCTURE ONLY (like everything under oracle/): the product path never imports this.
The image backbone, neck, depth net, BEV trunk, CenterPoint head, target assignment and detection loss are
the package's nn.Modules executed by torch on the CPU (plain PyTorch ops there: the HIP kernels only engage
on the GPU); get_geometry + binning, the lift (softmax (x) context) and voxel pooling forward / backward --
the reference's native / hot ops -- are the oracle's numpy + scalar-C restatements:
    lss_fpn.py:200-240,311-313 -> oracle.lss_geometry     :289-310 -> oracle.lss_lift
    voxel_pooling_ext (:48-59) -> oracle.bev_pool_fwd     VoxelPooling.backward (:64-79) -> oracle.bev_pool_bwd
"""
import numpy as np
import torch

import oracle


class _LiftSplatCPU(torch.autograd.Function):
    @staticmethod
    def forward(ctx, depth_feature, bins, D, C, nx, ny, nz):
        x = depth_feature.detach().numpy()
        lifted, prob = oracle.lss_lift(x, D, C)                         # [BN, D, fH, fW, C]
        BN = x.shape[0]
        feat = lifted.reshape(1, -1, C)
        out, pos = oracle.bev_pool_fwd(bins, feat, nx, ny, nz)           # [B, ny, nx, C]
        ctx.save_for_backward(depth_feature)
        ctx.aux = (prob, pos, D, C, lifted.shape)
        return torch.from_numpy(out).permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, gout):
        (depth_feature,) = ctx.saved_tensors
        prob, pos, D, C, lshape = ctx.aux
        gfeat = oracle.bev_pool_bwd(np.ascontiguousarray(gout.numpy()), pos)      # [B, N, C]
        g = gfeat.reshape(lshape)                                        # [BN, D, fH, fW, C]
        x = depth_feature.detach().numpy()
        ctxf = x[:, D:D + C]                                             # [BN, C, fH, fW]
        gprob = np.einsum("ndhwc,nchw->ndhw", g, ctxf, optimize=True)
        gctx = np.einsum("ndhwc,ndhw->nchw", g, prob, optimize=True)
        gz = prob * (gprob - (gprob * prob).sum(1, keepdims=True))       # softmax backward
        gx = np.zeros_like(x)
        gx[:, :D], gx[:, D:D + C] = gz, gctx
        return torch.from_numpy(gx), None, None, None, None, None, None


def build(seed=1234):
    """-> (model, batch): the package's BEVFusionCenterHead (camera only) on the CPU + a 1-camera batch."""
    from unidistill_amd import config as C, synthetic as syn
    from unidistill_amd.models import BEVFusionCenterHead
    torch.manual_seed(seed)
    model = BEVFusionCenterHead(C.model_cfg(lidar=False, camera=True)).train()
    g = syn.rng(seed)
    H, W = C.IMG_DIM
    imgs = torch.from_numpy(g.standard_normal((1, 1, 1, 3, H, W)).astype(np.float32))
    s2e, intr, ida, bda = syn.camera_rig(g, 1, 1)
    boxes, labels = syn.gt_boxes(g, 1, 40, 50)
    gt = torch.from_numpy(np.concatenate([boxes, labels[..., None] + 1.0], -1))
    return model, dict(imgs=imgs, s2e=s2e[:, 0], intr=intr[:, 0], ida=ida[:, 0], bda=bda, gt=gt)


def step(model, batch):
    """One forward + backward of the camera student on the CPU; returns the loss value."""
    enc = model.camera_encoder.backbone
    D, C = enc.depth_channels, enc.output_channels
    nx, ny, nz = enc._nxyz
    for p in model.parameters():
        p.grad = None
    feats = enc.get_cam_feats(batch["imgs"])[:, 0]
    depth_feature = enc.depth_net(feats.reshape(-1, *feats.shape[2:]))
    fr = enc.frustum.numpy()
    _, bins = oracle.lss_geometry(batch["s2e"], batch["intr"], batch["ida"], batch["bda"], fr[0, 0, :, 0],
                                  fr[0, :, 0, 1], fr[:, 0, 0, 2], enc.voxel_coord.numpy(), enc.voxel_size.numpy())
    bev = _LiftSplatCPU.apply(depth_feature, np.ascontiguousarray(bins.reshape(1, -1, 3)), D, C, nx, ny, nz)
    trunk, _ = model.bev_encoder(bev)
    ret = model.det_head(trunk, batch["gt"])
    loss, _ = model.det_head.dense_head.get_loss(ret)
    loss.backward()
    return float(loss)
