"""Fused target-assignment kernel (ud_assign_targets) vs the tensor-op formulation, which the reference
golden pins (tests/test_dense_head.py::test_assigner_targets_bit_exact)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("trial", range(5))
def test_fused_assignment_equals_tensor_ops(hip_lib, trial):
    from test_dense_head import _head, TASKS
    head = _head()
    asg = head.target_assigner
    g = torch.Generator().manual_seed(40 + trial)
    names = [n for t in TASKS for n in t["class_names"]]
    B, M = 3, 30
    gt = torch.zeros(B, M, 10)
    n_valid = [M, 9, 0]
    for b in range(B):
        n = n_valid[b]
        xy = (torch.rand(n, 2, generator=g) * 2 - 1) * (33.5 if trial % 2 else 31.0)       # some outside
        if trial == 3 and n:
            xy[: n // 2] = xy[0] + torch.randn(n // 2, 2, generator=g) * 0.6               # crowded
        gt[b, :n, 0:2] = xy
        gt[b, :n, 2] = torch.randn(n, generator=g)
        gt[b, :n, 3:6] = torch.rand(n, 3, generator=g) * 3 + 0.5
        gt[b, :n, 6] = (torch.rand(n, generator=g) * 2 - 1) * 6.0
        gt[b, :n, 7:9] = torch.randn(n, 2, generator=g)
        gt[b, :n, 9] = torch.randint(1, len(names) + 1, (n,), generator=g).float()
    if trial == 4:
        gt[0, 3, 3:6] = 0.0                     # zero-size box: log(0) = -inf in both paths
        gt[0, 5] = 0.0                          # an all-zero row in the middle stays "valid" (before the last)
    dev = gt.cuda()
    asg.fused = True
    a = asg.assign_targets(dev)
    asg.fused = False
    b_ = asg.assign_targets(dev)
    asg.fused = True
    for key in ("heatmap", "ind", "mask", "cat"):
        for t in a[key]:
            assert a[key][t].dtype == b_[key][t].dtype and torch.equal(a[key][t], b_[key][t]), (key, t)
    for t in a["box_encoding"]:
        x, y = a["box_encoding"][t], b_["box_encoding"][t]
        assert x.shape == y.shape
        same_inf = torch.isinf(x) == torch.isinf(y)
        assert same_inf.all()
        fin = ~torch.isinf(y)
        assert torch.allclose(x[fin], y[fin], rtol=1e-6, atol=1e-6), (t, (x[fin] - y[fin]).abs().max())
    assert torch.equal(a["_stacked"]["heatmap"], b_["_stacked"]["heatmap"])
