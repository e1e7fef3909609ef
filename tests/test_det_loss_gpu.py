"""Fused detection-loss kernels (ud_det_focal_*, ud_det_reg_*) vs the tensor-op formulation of
CenterHeadIouAware.get_loss, which itself is pinned to the reference by tests/test_dense_head.py."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _setup(seed, B=2, M=12):
    from test_dense_head import _head, TASKS
    torch.manual_seed(seed)
    head = _head().cuda()
    with torch.no_grad():
        head.auto_loss.params.copy_(torch.linspace(0.8, 1.3, 12))
        for p in head.tasks.parameters():             # spread the predictions (dims / rot / iou heads)
            p.add_(torch.randn_like(p) * 0.05)
    g = torch.Generator().manual_seed(seed)
    ncls = sum(len(t["class_names"]) for t in TASKS)
    gt = torch.zeros(B, M, 10)
    for b, n in enumerate([M - 2, 3][:B]):
        gt[b, :n, 0:2] = (torch.rand(n, 2, generator=g) * 2 - 1) * 29
        gt[b, :n, 2] = torch.randn(n, generator=g)
        gt[b, :n, 3:6] = torch.rand(n, 3, generator=g) * 3 + 0.4
        gt[b, :n, 6] = (torch.rand(n, generator=g) * 2 - 1) * 3.1
        gt[b, :n, 7:9] = torch.randn(n, 2, generator=g)
        gt[b, :n, 9] = torch.randint(1, ncls + 1, (n,), generator=g).float()
    feat = torch.randn(B, 24, 32, 32, generator=g)
    return head, gt.cuda(), feat.cuda()


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_fused_detection_loss_matches_tensor_ops(hip_lib, seed, lenient):
    head, gt, feat = _setup(seed)
    res = {}
    for fused in (False, True):
        head.fused_loss = fused
        head.train(); head.zero_grad()
        x = feat.clone().requires_grad_(True)
        ret = head(x, gt.clone())
        for enc in ret["box_encoding"].values():
            enc[torch.isinf(enc)] = 0
        loss, tb = head.get_loss(ret)
        probe = sum((pd["hm"] ** 2).mean() for pd in ret["multi_head_features"])   # gradient THROUGH prob too
        (loss + 0.3 * probe).backward()
        res[fused] = (loss.detach(), {k: v.detach().float().cpu() for k, v in tb.items()}, x.grad.clone(),
                      [p.grad.clone() for p in head.parameters() if p.grad is not None],
                      [pd["hm"].detach().clone() for pd in ret["multi_head_features"]])
    ref, got = res[False], res[True]
    np.testing.assert_allclose(got[0].cpu().numpy(), ref[0].cpu().numpy(), rtol=2e-5)
    for k in ref[1]:
        np.testing.assert_allclose(got[1][k].numpy(), ref[1][k].numpy(), rtol=5e-5, atol=1e-6, err_msg=k)
    for a, b in zip(got[4], ref[4]):
        np.testing.assert_allclose(a.cpu().numpy(), b.cpu().numpy(), rtol=1e-5, atol=1e-7)
    scale = float(ref[2].abs().max())
    np.testing.assert_allclose(got[2].cpu().numpy(), ref[2].cpu().numpy(), rtol=0, atol=2e-4 * scale)
    assert len(got[3]) == len(ref[3])
    gmax = max(float(b.abs().max()) for b in ref[3])     # biases in front of a BatchNorm have ~0 gradients
    for a, b in zip(got[3], ref[3]):
        np.testing.assert_allclose(a.cpu().numpy(), b.cpu().numpy(), rtol=0,
                                   atol=2e-4 * float(b.abs().max()) + 1e-6 * gmax)
