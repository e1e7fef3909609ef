"""GPU: end-to-end training steps (detector and distillation) run, produce finite losses and
update every trainable parameter; small shapes keep it quick."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _finite_grads(params):
    n = 0
    for p in params:
        if p.grad is not None:
            assert torch.isfinite(p.grad).all()
            n += 1
    return n


def test_lidar_detector_step():
    from unidistill_amd import train
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    step = train.DetectStep("lidar")
    tr = train.Trainer(step, device=dev)
    batch = train.synthetic_batch(dev, batch_size=1, with_imgs=False)
    before = [p.detach().clone() for p in tr.params[:5]]
    out = tr.step(batch)
    assert torch.isfinite(out["loss"])
    assert any(not torch.equal(a, b) for a, b in zip(before, tr.params[:5]))
    out2 = tr.step(batch)
    assert torch.isfinite(out2["loss"])


def test_camera_student_lidar_teacher_distill_step():
    from unidistill_amd import train
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    step = train.DistillStep("camera_exp_distill_lidar")
    tr = train.Trainer(step, device=dev)
    batch = train.synthetic_batch(dev, batch_size=1, ncam=6)
    out = tr.step(batch)
    assert torch.isfinite(out["loss"])
    for k in ("loss_feature", "loss_bev_rel", "loss_resp_cls", "loss_resp_reg", "loss_rpn"):
        assert torch.isfinite(out["tb"][k]), k
    assert all(p.grad is None for p in step.teacher_model.parameters())
    assert _finite_grads(step.model.parameters()) > 100
    # depth_net and the image backbone receive gradient through the fused lift+splat
    assert step.model.camera_encoder.backbone.depth_net[0].weight.grad.abs().sum() > 0
    assert step.model.camera_encoder.backbone.img_backbone.conv1.weight.grad.abs().sum() > 0
