"""Proposal layer (top-K decode + rotated NMS): the oracle's restatement vs the reference's own classes (golden,
CPU), oracle NMS / IoU self-checks (CPU); the device layer vs the same golden and vs the oracle at nuScenes
sizes, and the HIP NMS vs the oracle (GPU)."""
import numpy as np
import pytest
import torch

import oracle


_TASKS = [["car"], ["truck", "bus"], ["barrier"]]
_CFG = dict(dataset_name="nuscenes", class_names=_TASKS, post_center_limit_range=[-30.0, -30.0, -6.0, 30.0, 30.0, 6.0],
            score_threshold=0.1, pc_range=[-32.0, -32.0], out_size_factor=8, voxel_size=[0.25, 0.25], no_log=False,
            iou_aware_list=[0.65] * 3, nms_iou_threshold_train=0.8, nms_pre_max_size_train=60,
            nms_post_max_size_train=20, nms_iou_threshold_test=0.2, nms_pre_max_size_test=50,
            nms_post_max_size_test=12)
_KEYS = ("hm", "reg", "height", "dim", "rot", "vel", "iou")


def _oracle_layer(heads, cfg, phase):
    return oracle.proposal_layer(
        heads, cfg["class_names"], cfg["post_center_limit_range"], cfg["score_threshold"], cfg["pc_range"],
        cfg["out_size_factor"], cfg["voxel_size"], cfg["no_log"], cfg[f"nms_iou_threshold_{phase}"],
        cfg[f"nms_pre_max_size_{phase}"], cfg[f"nms_post_max_size_{phase}"], cfg["iou_aware_list"])


def test_oracle_proposal_layer_matches_reference(golden):
    """oracle.proposal_layer (numpy restatement + oracle NMS) == the reference's IouAwareGenProposals run on the
    same head tensors (golden made by tests/golden/make_goldens.py with the oracle bound to the missing NMS
    binary): pins everything around the NMS to the reference's own code."""
    gd = golden("proposals")
    heads = [{k: gd[f"in{t}_{k}"] for k in _KEYS} for t in range(3)]
    for phase in ("train", "test"):
        rois, scores, labels, counts = _oracle_layer(heads, _CFG, phase)
        np.testing.assert_allclose(rois, gd[f"{phase}_rois"], rtol=1e-6, atol=1e-6)
        np.testing.assert_allclose(scores, gd[f"{phase}_roi_scores"], rtol=1e-6, atol=1e-7)
        np.testing.assert_array_equal(labels, gd[f"{phase}_roi_labels"])
        for b in range(2):
            assert counts[b] == int(gd[f"{phase}_n{b}"][0])


@pytest.mark.gpu
@pytest.mark.parametrize("layout", ["nchw", "channels_last_slices"])
def test_proposal_layer_matches_reference_hip(hip_lib, golden, layout):
    """The device proposal layer (ud_proposal_layer: top-K decode, filters, HIP rotated NMS, roi packing) on the
    reference's golden: rois / scores 1e-6, labels and counts exact.  Second layout: the head tensors are
    channel slices of one packed channels-last map, read in place through their strides."""
    from unidistill_amd.layers.gen_proposals import IouAwareGenProposals
    gd = golden("proposals")
    prop = IouAwareGenProposals(**_CFG)
    dev = torch.device("cuda:0")
    heads = []
    for t in range(3):
        d = {k: torch.from_numpy(gd[f"in{t}_{k}"]).to(dev) for k in _KEYS}
        if layout != "nchw":
            packed = torch.cat([d[k] for k in _KEYS], 1).permute(0, 2, 3, 1).contiguous()      # [B,H,W,Ctot]
            c0, sl = 0, {}
            for k in _KEYS:
                c = d[k].shape[1]
                sl[k] = packed[..., c0:c0 + c].permute(0, 3, 1, 2)                             # strided view
                c0 += c
            d = sl
        heads.append(d)
    for phase in ("train", "test"):
        prop.train(phase == "train")
        res = prop.generate_predicted_boxes({"multi_head_features": heads}, {})
        np.testing.assert_allclose(res["rois"].cpu().numpy(), gd[f"{phase}_rois"], rtol=1e-6, atol=1e-6)
        np.testing.assert_allclose(res["roi_scores"].cpu().numpy(), gd[f"{phase}_roi_scores"], rtol=1e-6, atol=1e-7)
        np.testing.assert_array_equal(res["roi_labels"].cpu().numpy(), gd[f"{phase}_roi_labels"])
        for b, pd in enumerate(res["pred_dicts"]):
            n = int(gd[f"{phase}_n{b}"][0])
            assert pd["pred_boxes"].shape == (n, 9) and pd["pred_scores"].shape == (n,)
            np.testing.assert_array_equal(pd["pred_labels"].cpu().numpy(), gd[f"{phase}_roi_labels"][b, :n])


@pytest.mark.gpu
@pytest.mark.parametrize("phase,seed", [("train", 11), ("test", 12)])
def test_proposal_layer_full_size_vs_oracle(hip_lib, phase, seed):
    """nuScenes sizes (6 tasks, 180 x 180, K = 1500, post 80 / 100, B = 2): device layer == numpy restatement
    (labels / counts exact, boxes 1e-5), and no host synchronisation besides the count read."""
    from unidistill_amd.layers.gen_proposals import IouAwareGenProposals
    tasks = [["car"], ["truck", "construction_vehicle"], ["bus", "trailer"], ["barrier"],
             ["motorcycle", "bicycle"], ["pedestrian", "traffic_cone"]]
    c = dict(dataset_name="nuscenes", class_names=tasks,
             post_center_limit_range=[-61.2, -61.2, -10.0, 61.2, 61.2, 10.0], score_threshold=0.1,
             pc_range=[-54.0, -54.0], out_size_factor=8, voxel_size=[0.075, 0.075], no_log=False,
             iou_aware_list=[0.65] * 10, nms_iou_threshold_train=0.8, nms_pre_max_size_train=1500,
             nms_post_max_size_train=80, nms_iou_threshold_test=0.1, nms_pre_max_size_test=1500,
             nms_post_max_size_test=100)
    rng = np.random.default_rng(seed)
    B, H, W = 2, 180, 180
    heads = []
    for names in tasks:
        hm = rng.standard_normal((B, len(names), H, W)).astype(np.float32) * 1.5 - 2.19
        # cluster the top scores around a few centres so that NMS has something to suppress
        for b in range(B):
            for _ in range(25):
                cy, cx, k = rng.integers(5, H - 5), rng.integers(5, W - 5), rng.integers(0, len(names))
                hm[b, k, cy - 2:cy + 3, cx - 2:cx + 3] += 4.0
        heads.append({"hm": hm, "reg": rng.random((B, 2, H, W), np.float32),
                      "height": rng.standard_normal((B, 1, H, W)).astype(np.float32),
                      "dim": (rng.standard_normal((B, 3, H, W)) * 0.3 + 0.9).astype(np.float32),
                      "rot": rng.standard_normal((B, 2, H, W)).astype(np.float32),
                      "vel": rng.standard_normal((B, 2, H, W)).astype(np.float32),
                      "iou": rng.standard_normal((B, 1, H, W)).astype(np.float32)})
    ref_rois, ref_scores, ref_labels, ref_counts = _oracle_layer(heads, c, phase)
    prop = IouAwareGenProposals(**c).train(phase == "train")
    dev = torch.device("cuda:0")
    dheads = [{k: torch.from_numpy(v).to(dev) for k, v in d.items()} for d in heads]
    res = prop.generate_predicted_boxes({"multi_head_features": dheads}, {})
    assert [pd["pred_boxes"].shape[0] for pd in res["pred_dicts"]] == list(ref_counts)
    assert min(ref_counts) > 100                                   # the case is not degenerate
    np.testing.assert_array_equal(res["roi_labels"].cpu().numpy(), ref_labels)
    np.testing.assert_allclose(res["roi_scores"].cpu().numpy(), ref_scores, rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(res["rois"].cpu().numpy(), ref_rois, rtol=1e-5, atol=1e-5)
    # the layer itself never synchronises: with sync debugging on, only the .tolist() of the counts may
    prop._phase()
    torch.cuda.set_sync_debug_mode("error")
    try:
        offs = [0, 1, 3, 5, 6, 8]
        out = prop._run(dheads, offs, prop._alphas(6))
    finally:
        torch.cuda.set_sync_debug_mode("default")
    assert out[3].tolist() == list(ref_counts)


@pytest.mark.gpu
def test_proposal_layer_degenerate_maps(hip_lib):
    """Constant heat map (every score equal: ties resolved by ascending (class, pixel)), nothing above the
    score threshold, and K larger than the map."""
    from unidistill_amd.layers.gen_proposals import CenterPointGenProposals
    c = dict(_CFG)
    c.pop("iou_aware_list")
    c.update(dataset_name="kitti", class_names=[["car", "van"]], nms_pre_max_size_test=40, nms_post_max_size_test=10)
    B, H, W = 2, 6, 5                                            # 2 * 30 = 60 scores >= K = 40
    z = lambda ch: np.zeros((B, ch, H, W), np.float32)
    heads = [{"hm": z(2) + 1.0, "reg": z(2) + 0.5, "height": z(1), "dim": z(3), "rot": z(2) + 1.0}]
    heads[0]["hm"][1] = -9.0                                     # sample 1: sigmoid = 1.2e-4 < threshold
    ref = oracle.proposal_layer(heads, c["class_names"], c["post_center_limit_range"], 0.1, c["pc_range"], 8,
                                c["voxel_size"], False, 0.2, 40, 10, None, with_vel=False)
    prop = CenterPointGenProposals(**c).eval()
    dev = torch.device("cuda:0")
    res = prop.generate_predicted_boxes({"multi_head_features": [{k: torch.from_numpy(v).to(dev)
                                                                   for k, v in heads[0].items()}]}, {})
    assert res["rois"].shape == (B, 10, 7)
    np.testing.assert_array_equal(res["roi_labels"].cpu().numpy(), ref[2])
    np.testing.assert_allclose(res["rois"].cpu().numpy(), ref[0], rtol=1e-6, atol=1e-6)
    assert [pd["pred_boxes"].shape[0] for pd in res["pred_dicts"]] == list(ref[3]) and ref[3][1] == 0


def _rand_boxes(rng, n, spread=6.0):
    b = np.zeros((n, 7), np.float32)
    b[:, 0:2] = rng.uniform(-spread, spread, (n, 2))
    b[:, 2] = rng.uniform(-1, 1, n)
    b[:, 3:6] = rng.uniform(0.4, 4.5, (n, 3))
    b[:, 6] = rng.uniform(-3.3, 3.3, n)
    return b


def test_oracle_iou_against_independent_formulation():
    """Sutherland-Hodgman restatement (float) vs chord integration in double precision + closed forms."""
    rng = np.random.default_rng(3)
    b = _rand_boxes(rng, 50, 4.0)
    np.testing.assert_allclose(oracle.iou_bev(b, b), oracle.iou_bev_f64(b, b), rtol=0, atol=2e-5)
    x = np.array([[0, 0, 0, 2, 2, 1, 0]], np.float32)
    np.testing.assert_allclose(oracle.iou_bev(x, np.array([[1, 0, 0, 2, 2, 1, 0]], np.float32)), [[1 / 3]], atol=1e-6)
    np.testing.assert_allclose(oracle.iou_bev(x, np.array([[0, 0, 5, 2, 2, 1, np.pi / 2]], np.float32)), [[1.0]], atol=1e-5)
    sq = 2 * np.sqrt(2) - 2            # octagon: unit-half-width square vs itself rotated 45 deg
    inter = 8 * np.tan(np.pi / 8)
    np.testing.assert_allclose(oracle.iou_bev(x, np.array([[0, 0, 0, 2, 2, 1, np.pi / 4]], np.float32)),
                               [[inter / (8 - inter)]], atol=1e-5)
    assert sq > 0
    far = np.array([[50, 50, 0, 1, 1, 1, 0.3]], np.float32)
    assert oracle.iou_bev(x, far)[0, 0] == 0.0


def test_oracle_nms_properties():
    rng = np.random.default_rng(4)
    b = _rand_boxes(rng, 200)
    for thresh in (0.05, 0.3, 0.7):
        keep = oracle.nms_bev(b, thresh)
        assert keep[0] == 0 and np.all(np.diff(keep) > 0)
        iou = oracle.iou_bev(b[keep], b[keep])
        assert (iou - np.eye(len(keep)) <= thresh).all()           # survivors do not overlap
        dropped = np.setdiff1d(np.arange(len(b)), keep)
        for j in dropped:                                           # each dropped box had an earlier keeper
            earlier = keep[keep < j]
            assert (oracle.iou_bev(b[earlier], b[j:j + 1]) > thresh).any()
    assert len(oracle.nms_bev(b[:0], 0.5)) == 0


@pytest.mark.gpu
@pytest.mark.parametrize("n,thresh", [(1, 0.5), (63, 0.1), (64, 0.3), (500, 0.2), (1500, 0.1), (4100, 0.7)])
def test_hip_nms_equals_oracle(hip_lib, n, thresh):
    from unidistill_amd.ops import nms
    rng = np.random.default_rng(n)
    b = _rand_boxes(rng, n, spread=3.0 + n ** 0.5)
    iou = oracle.iou_bev(b[:min(n, 400)], b[:min(n, 400)])
    dev = torch.from_numpy(b).cuda()
    got = nms.boxes_iou_bev_gpu(dev[:400], dev[:400]).cpu().numpy()
    np.testing.assert_allclose(got, iou, rtol=0, atol=1e-4)      # device sinf/cosf vs libm: a few ulp on corners
    kept, count = nms._run(dev, thresh)
    ref = oracle.nms_bev(b, thresh)
    k = int(count.item())
    assert k == len(ref)
    np.testing.assert_array_equal(kept[:k].cpu().numpy(), ref)
    assert (kept[k:] == -1).all()
    keep = torch.zeros(n, dtype=torch.long)                        # reference-style entry point
    assert nms.nms_gpu(dev, keep, thresh) == k and np.array_equal(keep[:k].numpy(), ref)


@pytest.mark.gpu
def test_eval_forward_returns_boxes(hip_lib):
    """model.eval()(...) -> pred_dicts / rois like the reference's test path (center_head.py:142-146)."""
    from unidistill_amd import train
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    model = train.build_model("lidar").to(dev).eval()
    batch = train.synthetic_batch(dev, 2, with_imgs=False, with_points=True)
    with torch.no_grad():
        out = model([p for p in batch["points"]], None, None, None)
    assert set(out) >= {"pred_dicts", "rois", "roi_scores", "roi_labels"}
    assert len(out["pred_dicts"]) == 2 and out["rois"].shape == (2, 100 * 6, 9)
    for pd in out["pred_dicts"]:
        n = pd["pred_boxes"].shape[0]
        assert pd["pred_scores"].shape == (n,) and pd["pred_labels"].shape == (n,) and n <= 600
        if n:
            assert pd["pred_labels"].min() >= 1 and pd["pred_labels"].max() <= 10
