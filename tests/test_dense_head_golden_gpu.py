"""GPU: the HIP kernels of the dense head meet the REFERENCE golden directly (tests/golden/dense_head.npz,
produced by the reference's CenterHeadIouAware / FCOSAssigner / get_loss / BaseBEVBackbone on CPU):
  a14 ud_assign_targets      -> heat map / ind / mask / cat bit-exact, encodings 1e-6
  a15 ud_det_focal_* / ud_det_reg_*  -> loss, per-task terms, feature and AutomaticWeightedLoss gradients
  a12 BaseBEVBackbone on the GPU (fp32) and a13 packed head outputs.
(the CPU twins of these checks live in test_dense_head.py and are not part of the -m gpu run)"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _sd(g, prefix):
    return {k[len(prefix):]: torch.from_numpy(g[k]) for k in g.files if k.startswith(prefix)}


def test_hip_assigner_targets_vs_reference_golden(golden, hip_lib):
    from test_dense_head import _head
    g = golden("dense_head")
    head = _head().cuda()
    asg = head.target_assigner
    asg.fused = True                                   # ud_assign_targets
    tg = head.assign_targets(torch.from_numpy(g["head_gt"]).cuda())
    for t in range(3):
        enc = tg["box_encoding"][t]
        enc[torch.isinf(enc)] = 0
        np.testing.assert_array_equal(tg["heatmap"][t].cpu().numpy(), g[f"head_tgt{t}_heatmap"])
        np.testing.assert_array_equal(tg["ind"][t].cpu().numpy(), g[f"head_tgt{t}_ind"])
        np.testing.assert_array_equal(tg["mask"][t].cpu().numpy(), g[f"head_tgt{t}_mask"])
        np.testing.assert_array_equal(tg["cat"][t].cpu().numpy(), g[f"head_tgt{t}_cat"])
        np.testing.assert_allclose(enc.cpu().numpy(), g[f"head_tgt{t}_box_encoding"], rtol=1e-6, atol=1e-6)
        assert tg["ind"][t].dtype == torch.int64 and tg["mask"][t].dtype == torch.bool


@pytest.mark.parametrize("fused_loss", [True, False])
def test_hip_head_forward_loss_and_grads_vs_reference_golden(golden, hip_lib, fused_loss, lenient):
    from test_dense_head import _head
    g = golden("dense_head")
    head = _head()
    head.load_state_dict(_sd(g, "head_sd/"), strict=True)
    head = head.cuda().train()
    head.fused_loss = fused_loss                       # True: ud_det_focal_* / ud_det_reg_* kernels
    head.target_assigner.fused = True
    feat = torch.from_numpy(g["head_feat"]).cuda().requires_grad_(True)
    ret = head(feat, torch.from_numpy(g["head_gt"]).cuda())
    for enc in ret["box_encoding"].values():
        enc[torch.isinf(enc)] = 0
    loss, tb = head.get_loss(ret)
    np.testing.assert_allclose(loss.item(), float(g["head_loss"]), rtol=2e-5)
    for t in range(3):
        for hn, v in ret["multi_head_features"][t].items():
            np.testing.assert_allclose(v.detach().cpu().numpy(), g[f"head_out{t}_{hn}"], rtol=2e-4, atol=2e-5)
        ref = g[f"head_tb{t}"]
        got = [tb[f"task_{t}/loss"].item(), tb[f"task_{t}/hm_loss"].item(), tb[f"task_{t}/loc_loss"].item(),
               tb[f"task_{t}/box_loss"][0].item(), tb[f"task_{t}/box_loss"][9].item()]
        np.testing.assert_allclose(got, ref, rtol=1e-4, atol=1e-6)
    loss.backward()
    gref = g["head_feat_grad"]
    np.testing.assert_allclose(feat.grad.cpu().numpy(), gref, rtol=0, atol=1e-3 * float(np.abs(gref).max()))
    np.testing.assert_allclose(head.auto_loss.params.grad.cpu().numpy(), g["head_params_grad"], rtol=1e-4, atol=1e-7)


def test_trunk_on_gpu_vs_reference_golden(golden, hip_lib, lenient):
    from unidistill_amd.layers.bev import BaseBEVBackbone
    g = golden("dense_head")
    m = BaseBEVBackbone([2, 2], [1, 2], [8, 16], [1, 2], [12, 12], 6)
    m.load_state_dict(_sd(g, "trunk_sd/"), strict=True)
    m = m.cuda()
    x = torch.from_numpy(g["trunk_x"]).cuda()
    m.eval()
    with torch.no_grad():
        y, pyr = m(x)
    np.testing.assert_allclose(y.cpu().numpy(), g["trunk_y_eval"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(pyr["spatial_features_2x"].cpu().numpy(), g["trunk_pyr2"], rtol=1e-4, atol=1e-5)
    m.train()
    y, _ = m(x)
    np.testing.assert_allclose(y.detach().cpu().numpy(), g["trunk_y_train"], rtol=1e-4, atol=1e-5)
