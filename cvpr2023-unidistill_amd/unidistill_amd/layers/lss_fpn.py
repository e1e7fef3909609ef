"""Camera encoder: Lift-Splat-Shoot with the fused HIP lift+splat.

Mirror of LSSFPN (unidistill/layers/blocks_3d/mmdet3d/lss_fpn.py:85-368): same constructor
arguments, buffers (voxel_size / voxel_coord / voxel_num / frustum), sub-module names
(img_backbone, img_neck, depth_net) and forward signature.  What changes underneath:
  * get_geometry + binning: one HIP kernel over the frustum (ud_lss_geometry) instead of ~10
    batched 4x4 matmuls and two torch.inverse calls;
  * depth softmax (x) context, permute and voxel_pooling: ud_lss_splat_fwd pools
    depth_prob * context straight into the BEV grid -- the [B,6,112,16,44,256] tensor (484 MB,
    written twice per step in the reference) never exists; backward likewise.
``materialise=True`` switches to the reference's op boundary (lift -> voxel_pooling) for A/B.
"""
import os

import torch
from torch import nn

from .dense import Conv2d

from ..ops import bev_pool as _bp
from ..ops import lss as _lss
from .image import build_backbone, build_neck


class LSSFPN(nn.Module):
    def __init__(self, x_bound, y_bound, z_bound, d_bound, final_dim, downsample_factor,
                 output_channels, img_backbone_conf, img_neck_conf, depth_net_conf,
                 timestamp_net_conf=None, materialise=False, inverse="torch"):
        super().__init__()
        # "torch" (default = the reference's behaviour, index work is exact): torch.linalg.inv_ex on the device,
        # i.e. the reference's own ida_mat.inverse() / torch.inverse(intrin_mat) calls (lss_fpn.py:222,233) with
        # that backend's rounding -- two tiny solver launches, no host sync.  "exact": correctly rounded 4x4
        # inverses inside the matrix kernel (no solver launch; differs from the LAPACK LU in the last bit, which
        # can move a point sitting on a bin edge).
        assert inverse in ("exact", "torch")
        self.inverse = inverse
        self.downsample_factor = downsample_factor
        self.d_bound = d_bound
        self.final_dim = final_dim
        self.output_channels = output_channels
        self.materialise = materialise
        self.fuse_binning = True      # bins computed inside the splat's list-building kernel (ud_lss_splat_geom_fwd)
        bounds = [x_bound, y_bound, z_bound]
        self.register_buffer("voxel_size", torch.Tensor([r[2] for r in bounds]))
        self.register_buffer("voxel_coord", torch.Tensor([r[0] + r[2] / 2.0 for r in bounds]))
        self.register_buffer("voxel_num", torch.LongTensor([round((r[1] - r[0]) / r[2]) for r in bounds]))
        self._nxyz = tuple(int(round((r[1] - r[0]) / r[2])) for r in bounds)   # host copy: no .cuda()/sync
        self._lo, self._size = _lss.bin_origin_fp32([r[0] + r[2] / 2.0 for r in bounds],
                                                    [r[2] for r in bounds])
        self.register_buffer("frustum", self.create_frustum())
        self.depth_channels = self.frustum.shape[0]
        self.img_backbone = build_backbone(img_backbone_conf)
        self.img_neck = build_neck(img_neck_conf)
        self.depth_net = self._configure_depth_net(depth_net_conf)
        self.timestamp_net = None
        self.img_neck.init_weights()
        self.img_backbone.init_weights()
        # UD_GRAPH_IMAGE=1 / graph_image_branch: the shape-static image branch (backbone + neck + depth net, forward and backward)
        # replayed as two hipGraphs in training (ops/graphed.py); not a registered submodule: the state_dict keys stay the reference's
        object.__setattr__(self, "_image_graph", None)
        self.graph_image_branch = os.environ.get("UD_GRAPH_IMAGE", "0") == "1"

    def _configure_depth_net(self, conf):
        out_ch = self.depth_channels + self.output_channels
        if conf.get("num_res_layer", 0) != 0:
            raise NotImplementedError("depth_net with residual layers is not used by any experiment")
        return nn.Sequential(Conv2d(conf["in_channels"], out_ch, kernel_size=1))

    def create_frustum(self):
        """[D, fH, fW, 4] = (u, v, d, 1) in image pixels / metres (lss_fpn.py:173-198)."""
        H, W = self.final_dim
        fH, fW = H // self.downsample_factor, W // self.downsample_factor
        d = torch.arange(*self.d_bound, dtype=torch.float)
        u = torch.linspace(0, W - 1, fW, dtype=torch.float)
        v = torch.linspace(0, H - 1, fH, dtype=torch.float)
        D = d.numel()
        return torch.stack([u.view(1, 1, fW).expand(D, fH, fW), v.view(1, fH, 1).expand(D, fH, fW),
                            d.view(D, 1, 1).expand(D, fH, fW), torch.ones(D, fH, fW)], -1).contiguous()

    def _frustum_axes(self):
        fr = self.frustum
        return fr[0, 0, :, 0].contiguous(), fr[0, :, 0, 1].contiguous(), fr[:, 0, 0, 2].contiguous()

    def _frustum_desc(self, sensor2ego_mat, intrin_mat, ida_mat, bda_mat):
        """(mats, fu, fv, fd, lo, size, has_bda): everything the geometry of a frustum point needs (ops.lss.geometry's arguments)."""
        ai = ki = None
        if self.inverse == "torch":
            ai = torch.linalg.inv_ex(ida_mat.float()).inverse
            ki = torch.linalg.inv_ex(intrin_mat.float()).inverse
        mats = _lss.prepare_mats(sensor2ego_mat, intrin_mat, ida_mat, bda_mat, ai, ki)
        fu, fv, fd = self._frustum_axes()
        return mats, fu, fv, fd, self._lo, self._size, bda_mat is not None

    def get_geometry_bins(self, sensor2ego_mat, intrin_mat, ida_mat, bda_mat, want_geom=False):
        """-> bins i32[B, N, 3] (and optionally ego coordinates f32[B,ncam,D,fH,fW,3])."""
        B, ncam = sensor2ego_mat.shape[:2]
        mats, fu, fv, fd, lo, size, has_bda = self._frustum_desc(sensor2ego_mat, intrin_mat, ida_mat, bda_mat)
        return _lss.geometry(mats, fu, fv, fd, B, ncam, lo, size, has_bda=has_bda, want_geom=want_geom)

    def get_geometry(self, sensor2ego_mat, intrin_mat, ida_mat, bda_mat):
        return self.get_geometry_bins(sensor2ego_mat, intrin_mat, ida_mat, bda_mat, True)[1]

    def _backbone_neck(self, x):
        w = next((p for p in self.img_backbone.parameters() if p.dim() == 4), None)   # the stem convolution
        any_layout = getattr(self.img_backbone, "stem_takes_any_layout", None)
        if any_layout is not None and any_layout(x):
            pass                                                       # the HIP stem reads the images through their strides
        elif w is not None and w.is_contiguous(memory_format=torch.channels_last) and not w.is_contiguous():
            x = x.contiguous(memory_format=torch.channels_last)        # NHWC model -> NHWC input
        return self.img_neck(self.img_backbone(x))[0]

    def get_cam_feats(self, imgs):
        B, S, N, C, H, W = imgs.shape
        f = self._backbone_neck(imgs.reshape(B * S * N, C, H, W))
        return f.reshape(B, S, N, f.shape[1], f.shape[2], f.shape[3])

    def _image_branch(self, x):
        """images [B * ncam, 3, H, W] -> depth feature [B * ncam, D + C, fH, fW]: get_cam_feats + depth_net of one sweep."""
        return self.depth_net(self._backbone_neck(x))

    def _depth_feature(self, sweep_imgs):
        B, S, ncam = sweep_imgs.shape[:3]
        if self.graph_image_branch and S == 1 and self.training and torch.is_grad_enabled() and sweep_imgs.is_cuda:
            from ..ops.graphed import GraphedModule
            if self._image_graph is None:
                object.__setattr__(self, "_image_graph", GraphedModule("image_branch", self._image_branch,
                                                                      [self.img_backbone, self.img_neck, self.depth_net]))
            C, H, W = sweep_imgs.shape[3:]
            return self._image_graph(sweep_imgs.reshape(B * S * ncam, C, H, W))
        feats = self.get_cam_feats(sweep_imgs)[:, 0]
        return self.depth_net(feats.reshape(B * ncam, *feats.shape[2:]))

    def _forward_single_sweep(self, sweep_index, sweep_imgs, mats_dict, is_return_depth=False):
        B, S, ncam = sweep_imgs.shape[:3]
        depth_feature = self._depth_feature(sweep_imgs)
        D, C = self.depth_channels, self.output_channels
        geo = (mats_dict["sensor2ego_mats"][:, sweep_index], mats_dict["intrin_mats"][:, sweep_index],
               mats_dict["ida_mats"][:, sweep_index], mats_dict.get("bda_mat", None))
        nx, ny, nz = self._nxyz
        if self.materialise:
            bins, _ = self.get_geometry_bins(*geo)
            lifted = _lss.lift(depth_feature, D, C)                      # [B*ncam, D, fH, fW, C]
            bev = _bp.voxel_pooling(bins, lifted.reshape(B, -1, C), (nx, ny, nz))
        elif self.fuse_binning:
            # the bins are computed inside the splat's list-building kernel (same arithmetic, lss_geom.h): never written
            bev = _lss.lift_splat(depth_feature, self._frustum_desc(*geo), B, ncam, D, C, nx, ny, nz)
        else:
            bins, _ = self.get_geometry_bins(*geo)
            bev = _lss.lift_splat(depth_feature, bins, B, ncam, D, C, nx, ny, nz)
        if is_return_depth:
            return bev, depth_feature[:, :D].softmax(1)
        return bev

    def forward(self, sweep_imgs, mats_dict, timestamps=None, is_return_depth=False):
        """sweep_imgs f32[B, num_sweeps, num_cams, 3, H, W] -> BEV map [B, C*num_sweeps, ny, nx]
        (a channels-last view: the splat writes NHWC and returns its NCHW permutation)."""
        S = sweep_imgs.shape[1]
        key = self._forward_single_sweep(0, sweep_imgs[:, 0:1], mats_dict, is_return_depth)
        if S == 1:
            return key
        maps = [key[0] if is_return_depth else key]
        for s in range(1, S):
            with torch.no_grad():
                maps.append(self._forward_single_sweep(s, sweep_imgs[:, s:s + 1], mats_dict, False))
        out = torch.cat(maps, 1)
        return (out, key[1]) if is_return_depth else out
