"""Proposal layer (top-K decode + rotated NMS): mirror vs the reference's own classes (golden), oracle
NMS / IoU self-checks (CPU), and the HIP NMS vs the oracle (GPU)."""
import numpy as np
import pytest
import torch

import oracle


def _oracle_nms_fn(boxes, scores, thresh, pre, post):
    order = scores.sort(0, descending=True)[1]
    if pre is not None:
        order = order[:pre]
    kept = oracle.nms_bev(boxes[order][:, :7].detach().cpu().numpy(), float(thresh))
    sel = order[torch.from_numpy(kept).to(order.device)]
    return sel[:post] if post is not None else sel


def test_proposal_layer_matches_reference(golden):
    """layers/gen_proposals.IouAwareGenProposals == the reference class run on the same head tensors
    (golden made by tests/golden/make_goldens.py with the oracle bound to the missing NMS binary)."""
    from unidistill_amd.layers.gen_proposals import IouAwareGenProposals
    gd = golden("proposals")
    tasks = [["car"], ["truck", "bus"], ["barrier"]]
    prop = IouAwareGenProposals(
        dataset_name="nuscenes", class_names=tasks, post_center_limit_range=[-30.0, -30.0, -6.0, 30.0, 30.0, 6.0],
        score_threshold=0.1, pc_range=[-32.0, -32.0], out_size_factor=8, voxel_size=[0.25, 0.25], no_log=False,
        iou_aware_list=[0.65] * 3, nms_iou_threshold_train=0.8, nms_pre_max_size_train=60,
        nms_post_max_size_train=20, nms_iou_threshold_test=0.2, nms_pre_max_size_test=50,
        nms_post_max_size_test=12, nms_fn=_oracle_nms_fn)
    heads = [{k: torch.from_numpy(gd[f"in{t}_{k}"]) for k in ("hm", "reg", "height", "dim", "rot", "vel", "iou")}
             for t in range(3)]
    for phase in ("train", "test"):
        prop.train(phase == "train")
        res = prop.generate_predicted_boxes({"multi_head_features": heads}, {})
        np.testing.assert_allclose(res["rois"].numpy(), gd[f"{phase}_rois"], rtol=1e-6, atol=1e-6)
        np.testing.assert_allclose(res["roi_scores"].numpy(), gd[f"{phase}_roi_scores"], rtol=1e-6, atol=1e-7)
        np.testing.assert_array_equal(res["roi_labels"].numpy(), gd[f"{phase}_roi_labels"])
        for b, pd in enumerate(res["pred_dicts"]):
            assert pd["pred_boxes"].shape[0] == int(gd[f"{phase}_n{b}"][0])


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
