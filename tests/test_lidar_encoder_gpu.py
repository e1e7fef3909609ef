"""GPU: the full LiDAR branch (voxelize -> VFE -> sparse backbone -> BEV) vs a dense restatement
built from the oracle's voxels and torch.nn.functional.conv3d on a small grid."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import oracle

pytestmark = pytest.mark.gpu


def _dense_reference(model, feats, coords, shape, B):
    """Dense evaluation of the same network: convs as conv3d, sparsity enforced by site masks."""
    bb = model
    x = torch.from_numpy(oracle.sparse_to_dense(feats, coords, (B,) + shape)).cuda()
    occ = torch.zeros((B, 1) + shape, device="cuda")
    occ[coords[:, 0], 0, coords[:, 1], coords[:, 2], coords[:, 3]] = 1

    def wt(conv):
        return conv.weight.permute(0, 4, 1, 2, 3).contiguous()

    def bn(m, t, mask):
        # BatchNorm1d over active rows only (eval mode -> running stats)
        y = (t - m.running_mean.view(1, -1, 1, 1, 1)) / torch.sqrt(m.running_var.view(1, -1, 1, 1, 1) + m.eps)
        return (y * m.weight.view(1, -1, 1, 1, 1) + m.bias.view(1, -1, 1, 1, 1)) * mask

    def subm(conv, t, mask):
        return F.conv3d(t, wt(conv), conv.bias, padding=1) * mask

    def seq(s, t, mask):
        return torch.relu(bn(s[1], subm(s[0], t, mask), mask))

    def block(b, t, mask):
        y = torch.relu(bn(b.bn1, subm(b.conv1, t, mask), mask))
        y = bn(b.bn2, subm(b.conv2, y, mask), mask)
        return torch.relu(y + t) * mask

    def down(s, t, mask):
        c = s[0]
        nm = (F.conv3d(mask, torch.ones((1, 1) + c.kernel_size, device="cuda"), stride=c.stride,
                       padding=c.padding) > 0).float()
        y = F.conv3d(t, wt(c), None, stride=c.stride, padding=c.padding) * nm
        return torch.relu(bn(s[1], y, nm)), nm

    t = seq(bb.conv_input, x, occ)
    for b in bb.conv1:
        t = block(b, t, occ)
    m = occ
    for stage in (bb.conv2, bb.conv3, bb.conv4):
        t, m = down(stage[0], t, m)
        t = block(stage[1], t, m)
        t = block(stage[2], t, m)
    t, m = down(bb.conv_out, t, m)
    n, c, d, h, w = t.shape
    return t.reshape(n, c * d, h, w)


def test_lidar_encoder_matches_dense_restatement():
    from unidistill_amd.layers.lidar import LidarEncoder
    torch.manual_seed(0)
    # small world: 64 x 64 x 8 voxels -> sparse shape (9, 64, 64) -> BEV 8 x 8, z 9->5->3->1->... needs >= 41
    cfg = dict(voxel_size=[0.5, 0.5, 0.25], point_cloud_range=[-16.0, -16.0, -5.0, 16.0, 16.0, 5.0],
               grid_size=[64, 64, 40], max_num_points=10, max_voxels=(20000, 20000),
               src_num_point_features=5, use_num_point_features=5, map_to_bev_num_features=256)
    enc = LidarEncoder(cfg).cuda().eval()
    # non-trivial BN statistics
    for mod in enc.modules():
        if isinstance(mod, torch.nn.BatchNorm1d):
            mod.running_mean.normal_(0, 0.1)
            mod.running_var.uniform_(0.5, 1.5)
            mod.weight.data.uniform_(0.5, 1.5)
            mod.bias.data.normal_(0, 0.1)
    rng = np.random.default_rng(1)
    B, N = 2, 3000
    pts = np.concatenate([rng.uniform(-15.9, 15.9, (B, N, 2)), rng.uniform(-4.9, 4.9, (B, N, 1)),
                          rng.uniform(0, 1, (B, N, 2))], -1).astype(np.float32)
    with torch.no_grad():
        bev = enc([torch.from_numpy(pts[b]).cuda() for b in range(B)])
    assert bev.shape == (B, 256, 8, 8)
    ref_v = oracle.voxelize(pts, cfg["voxel_size"], cfg["point_cloud_range"], 10, 20000)
    with torch.no_grad():
        ref = _dense_reference(enc.backbone_3d, ref_v["mean"], ref_v["coords"], (41, 64, 64), B)
    np.testing.assert_allclose(bev.cpu().numpy(), ref.cpu().numpy(), rtol=2e-3, atol=2e-4)


def test_backbone_state_dict_keys_match_reference_names():
    from unidistill_amd.layers.lidar import VoxelResBackBone8x
    m = VoxelResBackBone8x(5, [1440, 1440, 40])
    keys = set(m.state_dict().keys())
    for k in ("conv_input.0.weight", "conv_input.1.running_mean", "conv1.0.conv1.weight",
              "conv1.0.conv1.bias", "conv1.1.bn2.weight", "conv2.0.0.weight", "conv2.0.1.bias",
              "conv2.2.conv2.bias", "conv4.0.0.weight", "conv_out.0.weight", "conv_out.1.num_batches_tracked"):
        assert k in keys, k
    assert tuple(m.state_dict()["conv4.0.0.weight"].shape) == (128, 3, 3, 3, 64)
    assert tuple(m.state_dict()["conv_out.0.weight"].shape) == (128, 3, 1, 1, 128)
    assert list(m.sparse_shape) == [41, 1440, 1440]
    n_params = sum(p.numel() for p in m.parameters())
    assert abs(n_params - 2.69e6) < 0.05e6      # SURVEY: sparse encoder 2.69 M params


def test_lidar_encoder_backward_runs_full_size():
    """cfg-2 size: 30k-pt cloud, real grid; loss.backward() reaches every parameter."""
    from unidistill_amd.layers.lidar import LidarEncoder
    from unidistill_amd import synthetic as syn
    cfg = dict(voxel_size=list(syn.VOXEL_SIZE), point_cloud_range=list(syn.POINT_CLOUD_RANGE),
               grid_size=list(syn.GRID_SIZE), max_num_points=10, max_voxels=(120000, 160000),
               src_num_point_features=5, use_num_point_features=5, map_to_bev_num_features=256)
    enc = LidarEncoder(cfg).cuda().train()
    pts = torch.from_numpy(syn.lidar_cloud(syn.rng(), 30000, 1)).cuda()
    bev = enc([pts])
    assert bev.shape == (1, 256, 180, 180)
    bev.square().mean().backward()
    for n, p in enc.named_parameters():
        assert p.grad is not None and torch.isfinite(p.grad).all(), n


def _sync_count(fn):
    """Number of blocking host<->device synchronisations fn() triggers (torch's sync debug mode, warn level)."""
    import warnings
    torch.cuda.synchronize()
    torch.cuda.set_sync_debug_mode(1)
    try:
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            out = fn()
    finally:
        torch.cuda.set_sync_debug_mode(0)
    return out, sum("synchroniz" in str(x.message).lower() for x in w)


@pytest.mark.parametrize("B,N", [(1, 30000), (4, 30000), (2, 250000)])
def test_prepare_reads_the_host_once_and_matches_the_per_level_path(B, N):
    """voxel count + overflow word + the four level sizes in ONE device->host read; sites, rulebooks and
    the BEV map identical (bit for bit) to the read-per-level path."""
    from unidistill_amd.layers.lidar import LidarEncoder
    from unidistill_amd.ops import spconv as sp
    from unidistill_amd import synthetic as syn
    cfg = dict(voxel_size=list(syn.VOXEL_SIZE), point_cloud_range=list(syn.POINT_CLOUD_RANGE),
               grid_size=list(syn.GRID_SIZE), max_num_points=10, max_voxels=(120000, 160000),
               src_num_point_features=5, use_num_point_features=5, map_to_bev_num_features=256)
    torch.manual_seed(0)
    enc = LidarEncoder(cfg).cuda().eval()
    g = syn.rng()
    pts = [torch.from_numpy(syn.lidar_cloud(g, N, 1)).cuda() for _ in range(B)]
    n = min(p.shape[0] for p in pts)
    pts = [p[:n].contiguous() for p in pts]            # equal lengths: the collate_fn case
    probe, n_probe = _sync_count(lambda: torch.ones(4, device="cuda").cpu())
    enc.one_read = True
    xa, n_one = _sync_count(lambda: enc.prepare(pts))
    enc.one_read = False
    xb, n_lvl = _sync_count(lambda: enc.prepare(pts))
    if n_probe:                                        # sync debug mode reports on this build
        assert n_one == 1, n_one
        assert n_lvl >= 5, n_lvl
    assert torch.equal(xa.indices, xb.indices) and torch.equal(xa.features, xb.features)
    sa, sb = xa._sites, xb._sites
    for m in enc.backbone_3d.modules():
        if isinstance(m, sp.SparseConv3d) and not m.subm and not m.inverse:
            (oa, na, ia), (ob, nb, ib) = (s.down(m.kernel_size, m.stride, m.padding) for s in (sa, sb))
            assert torch.equal(oa.indices, ob.indices) and torch.equal(na, nb) and torch.equal(ia, ib)
            assert torch.equal(oa.subm_rulebook((3, 3, 3)), ob.subm_rulebook((3, 3, 3)))
            sa, sb = oa, ob
    with torch.no_grad():
        ya, n_fwd = _sync_count(lambda: enc(pts, prepared=xa))
        yb = enc(pts, prepared=xb)
    if n_probe:
        assert n_fwd == 0, n_fwd
    assert torch.equal(ya, yb)
