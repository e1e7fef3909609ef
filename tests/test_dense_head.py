"""CPU: dense BEV trunk, CenterHeadIouAware, FCOSAssigner and the detection loss vs golden
vectors captured from the reference modules (state_dicts load unchanged -> key compatibility)."""
import numpy as np
import torch

from unidistill_amd.layers.bev import BaseBEVBackbone
from unidistill_amd.layers import center_head as ch

TASKS = [dict(num_class=1, class_names=["car"]), dict(num_class=2, class_names=["truck", "bus"]),
         dict(num_class=1, class_names=["barrier"])]


def _sd(g, prefix):
    return {k[len(prefix):]: torch.from_numpy(g[k]) for k in g.files if k.startswith(prefix)}


def test_trunk_matches_reference(golden):
    g = golden("dense_head")
    m = BaseBEVBackbone([2, 2], [1, 2], [8, 16], [1, 2], [12, 12], 6)
    missing = m.load_state_dict(_sd(g, "trunk_sd/"), strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    x = torch.from_numpy(g["trunk_x"])
    m.eval()
    with torch.no_grad():
        y, pyr = m(x)
    np.testing.assert_allclose(y.numpy(), g["trunk_y_eval"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(pyr["spatial_features_2x"].numpy(), g["trunk_pyr2"], rtol=1e-5, atol=1e-6)
    m.train()
    y, _ = m(x)
    np.testing.assert_allclose(y.detach().numpy(), g["trunk_y_train"], rtol=1e-4, atol=1e-5)


def _head():
    names = [n for t in TASKS for n in t["class_names"]]
    pc_range, voxel = [-32.0, -32.0, -5.0, 32.0, 32.0, 3.0], [0.25, 0.25, 0.2]
    assigner = ch.FCOSAssigner(out_size_factor=8, tasks=TASKS, dense_reg=1, gaussian_overlap=0.1,
                               max_objs=200, min_radius=2, mapping={n: i + 1 for i, n in enumerate(names)},
                               grid_size=[256, 256, 40], pc_range=pc_range[:2], voxel_size=voxel[:2],
                               assign_topk=9, with_velocity=True)
    return ch.CenterHeadIouAware(
        dataset_name="nuscenes", tasks=TASKS, target_assigner=assigner, proposal_layer=None,
        out_size_factor=8, input_channels=24, grid_size=[256, 256, 40], point_cloud_range=pc_range,
        code_weights=[1.0] * 8 + [0.2, 0.2], loc_weight=0.25, iou_weight=5.0, share_conv_channel=16,
        common_heads={"iou": [1, 2], "reg": [2, 2], "height": [1, 2], "dim": [3, 2], "rot": [2, 2], "vel": [2, 2]},
        voxel_size_xy=voxel[:2])


def test_assigner_targets_bit_exact(golden):
    g = golden("dense_head")
    head = _head()
    tg = head.assign_targets(torch.from_numpy(g["head_gt"]))
    for t in range(3):
        enc = tg["box_encoding"][t]
        enc[torch.isinf(enc)] = 0
        np.testing.assert_array_equal(tg["heatmap"][t].numpy(), g[f"head_tgt{t}_heatmap"])
        np.testing.assert_array_equal(tg["ind"][t].numpy(), g[f"head_tgt{t}_ind"])
        np.testing.assert_array_equal(tg["mask"][t].numpy(), g[f"head_tgt{t}_mask"])
        np.testing.assert_array_equal(tg["cat"][t].numpy(), g[f"head_tgt{t}_cat"])
        np.testing.assert_allclose(enc.numpy(), g[f"head_tgt{t}_box_encoding"], rtol=1e-6, atol=1e-6)
        assert tg["heatmap"][t].dtype == torch.float32 and tg["ind"][t].dtype == torch.int64
        assert tg["mask"][t].dtype == torch.bool


def test_head_forward_loss_and_grads(golden):
    g = golden("dense_head")
    head = _head()
    res = head.load_state_dict(_sd(g, "head_sd/"), strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    head.train()
    feat = torch.from_numpy(g["head_feat"]).requires_grad_(True)
    ret = head(feat, torch.from_numpy(g["head_gt"]))
    for enc in ret["box_encoding"].values():
        enc[torch.isinf(enc)] = 0
    loss, tb = head.get_loss(ret)
    np.testing.assert_allclose(loss.item(), float(g["head_loss"]), rtol=1e-5)
    for t in range(3):
        for hn, v in ret["multi_head_features"][t].items():
            np.testing.assert_allclose(v.detach().numpy(), g[f"head_out{t}_{hn}"], rtol=1e-4, atol=1e-5)
        ref = g[f"head_tb{t}"]
        got = [tb[f"task_{t}/loss"].item(), tb[f"task_{t}/hm_loss"].item(), tb[f"task_{t}/loc_loss"].item(),
               tb[f"task_{t}/box_loss"][0].item(), tb[f"task_{t}/box_loss"][9].item()]
        np.testing.assert_allclose(got, ref, rtol=1e-4, atol=1e-6)
    loss.backward()
    np.testing.assert_allclose(feat.grad.numpy(), g["head_feat_grad"], rtol=1e-3, atol=1e-6)
    np.testing.assert_allclose(head.auto_loss.params.grad.numpy(), g["head_params_grad"], rtol=1e-4, atol=1e-7)


def test_packed_heads_speak_reference_state_dict(golden):
    """The packed 2-conv head saves / loads the reference's per-head key names and shapes, and is
    numerically the per-head module stack."""
    g = golden("dense_head")
    ref_sd = _sd(g, "head_sd/")
    packed, plain = _head(), None
    packed.load_state_dict(ref_sd, strict=True)
    out_sd = packed.state_dict()
    assert set(out_sd.keys()) == set(ref_sd.keys())
    for k, v in ref_sd.items():
        assert tuple(out_sd[k].shape) == tuple(v.shape), k
        if v.dtype.is_floating_point:
            np.testing.assert_array_equal(out_sd[k].numpy(), v.numpy())
    from unidistill_amd.layers import center_head as chm
    names = [n for t in TASKS for n in t["class_names"]]
    plain = chm.CenterHeadIouAware(
        dataset_name="nuscenes", tasks=TASKS, target_assigner=packed.target_assigner, proposal_layer=None,
        out_size_factor=8, input_channels=24, grid_size=[256, 256, 40],
        point_cloud_range=[-32.0, -32.0, -5.0, 32.0, 32.0, 3.0], code_weights=[1.0] * 8 + [0.2, 0.2],
        loc_weight=0.25, iou_weight=5.0, share_conv_channel=16,
        common_heads={"iou": [1, 2], "reg": [2, 2], "height": [1, 2], "dim": [3, 2], "rot": [2, 2], "vel": [2, 2]},
        voxel_size_xy=[0.25, 0.25], packed_heads=False)
    plain.load_state_dict(ref_sd, strict=True)
    x = torch.from_numpy(g["head_feat"])
    packed.eval(); plain.eval()
    with torch.no_grad():
        a = packed(x)["multi_head_features"]
        b = plain(x)["multi_head_features"]
    for t in range(3):
        for k in b[t]:
            np.testing.assert_allclose(a[t][k].numpy(), b[t][k].numpy(), rtol=1e-4, atol=1e-5)


def test_nearest_bev_iou(golden):
    g = golden("dense_head")
    a, b = torch.from_numpy(g["iou_a"]), torch.from_numpy(g["iou_b"])
    np.testing.assert_allclose(ch.boxes3d_nearest_bev_iou(a, b).numpy(), g["iou_bev"], rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(ch.nearest_bev_iou_pairwise(a, b).numpy(), np.diag(g["iou_bev"]), rtol=1e-5, atol=1e-7)


def test_param_counts_match_survey():
    """BEV trunk 4.58 M, head 1.90 M params at the real widths."""
    trunk = BaseBEVBackbone([5, 5], [1, 2], [128, 256], [1, 2], [256, 256], 256)
    assert abs(sum(p.numel() for p in trunk.parameters()) - 4.58e6) < 0.02e6


def test_windowed_assignment_equals_dense_formulation():
    """The 9x9-window target assignment == the all-anchors formulation, incl. boxes at / beyond the
    range border, many boxes per task and empty tasks (bit-exact integer targets, equal encodings)."""
    import torch
    head = _head()
    asg = head.target_assigner
    g = torch.Generator().manual_seed(11)
    names = [n for t in TASKS for n in t["class_names"]]
    for trial in range(4):
        B, M = 3, 30
        gt = torch.zeros(B, M, 10)
        n_valid = [M, 7, 0][:B]
        for b in range(B):
            n = n_valid[b]
            xy = (torch.rand(n, 2, generator=g) * 2 - 1) * (33.5 if trial % 2 else 31.0)   # some outside
            if trial == 3 and n:
                xy[: n // 2] = xy[0] + torch.randn(n // 2, 2, generator=g) * 0.6           # crowded
            gt[b, :n, 0:2] = xy
            gt[b, :n, 2] = torch.randn(n, generator=g)
            gt[b, :n, 3:6] = torch.rand(n, 3, generator=g) * 3 + 0.5
            gt[b, :n, 6] = (torch.rand(n, generator=g) * 2 - 1) * 3.1
            gt[b, :n, 7:9] = torch.randn(n, 2, generator=g)
            gt[b, :n, 9] = torch.randint(1, len(names) + 1, (n,), generator=g).float()
        asg.windowed = True
        a = asg.assign_targets(gt)
        asg.windowed = False
        b_ = asg.assign_targets(gt)
        asg.windowed = True
        for key in ("heatmap", "ind", "mask", "cat", "box_encoding"):
            for t in a[key]:
                assert torch.equal(a[key][t], b_[key][t]), (trial, key, t)
