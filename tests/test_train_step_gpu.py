"""GPU: end-to-end training steps (detector and distillation) run, produce finite losses and
update every trainable parameter, for every BASELINE.json configuration (cfg 1-5); run-to-run / two-stream
agreement lives in test_step_determinism_gpu.py, numeric parity of the composed step in test_model_step_gpu.py."""
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
    tr = train.Trainer(step, device=dev, channels_last=True)     # the product layout: UD_STRICT holds (no library fall-through)
    batch = train.synthetic_batch(dev, batch_size=1, ncam=6)
    out = tr.step(batch)
    assert torch.isfinite(out["loss"])
    for k in ("loss_feature", "loss_bev_rel", "loss_resp_cls", "loss_resp_reg", "loss_rpn"):
        assert torch.isfinite(out["tb"][k]), k
    assert all(p.grad is None for p in step.teacher_model.parameters())
    assert _finite_grads(step.model.parameters()) > 100
    # depth_net and the image backbone receive gradient through the fused lift+splat; the stem is frozen
    # (frozen_stages=0, centerhead_fusion_exp.py:24-31 -> mmdet ResNet._freeze_stages)
    bb = step.model.camera_encoder.backbone.img_backbone
    assert step.model.camera_encoder.backbone.depth_net[0].weight.grad.abs().sum() > 0
    assert bb.layer1[0].conv1.weight.grad.abs().sum() > 0
    assert bb.conv1.weight.grad is None and bb.bn1.weight.grad is None and not bb.bn1.training


def _distill_step_checks(workload, batch_size, sweeps, ac, ncam=6):
    from unidistill_amd import train
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    step = train.DistillStep(workload)
    tr = train.Trainer(step, device=dev, autocast_dtype=ac, channels_last=True)
    batch = train.synthetic_batch(dev, batch_size=batch_size, ncam=ncam, sweeps=sweeps)
    before = {n: p.detach().clone() for n, p in step.model.named_parameters() if p.requires_grad}
    out = tr.step(batch)
    assert torch.isfinite(out["loss"])
    for k in ("loss_feature", "loss_bev_rel", "loss_resp_cls", "loss_resp_reg", "loss_rpn"):
        assert torch.isfinite(out["tb"][k]) and float(out["tb"][k]) > 0, k
    assert all(p.grad is None for p in step.teacher_model.parameters())
    assert not step.teacher_model.training and step.model.training
    n_moved = sum(int(not torch.equal(before[n], p.detach())) for n, p in step.model.named_parameters()
                  if p.requires_grad)
    assert n_moved >= 0.95 * len(before), (n_moved, len(before))
    assert _finite_grads(step.model.parameters()) == len(before)
    out2 = tr.step(batch)
    assert torch.isfinite(out2["loss"])
    return step, out


def test_cfg4_lidar_student_fusion_teacher_three_loss_step():
    """BASELINE configs[3]: LiDAR student + fusion teacher, full 3-loss distillation at its stated batch of 4 per GPU,
    fp32 (the reference's arithmetic)."""
    step, out = _distill_step_checks("lidar_exp_distill_fusion", batch_size=4, sweeps=1, ac=None)
    assert step.teacher_model.fusion_encoder is not None and step.teacher_model.camera_encoder is not None
    assert step.model.camera_encoder is None and step.model.lidar_encoder is not None
    e = step.exp
    assert (e["feat"], e["rel"], e["resp"]) == (10.0, 1.0, 10.0)       # lidar_exp_distill_fusion.py loss weights


def test_lidar_student_camera_teacher_step():
    """The fourth distillation experiment (..._lidar_exp_distill_camera.py:404,504-509): LiDAR student, camera
    teacher, loss weights 10 / 5 / 1."""
    step, out = _distill_step_checks("lidar_exp_distill_camera", batch_size=2, sweeps=1, ac=None)
    assert step.teacher_model.camera_encoder is not None and step.teacher_model.lidar_encoder is None
    assert step.model.camera_encoder is None and step.model.lidar_encoder is not None
    e = step.exp
    assert (e["feat"], e["rel"], e["resp"], e["clamp"]) == (10.0, 5.0, 1.0, 1e-4)


def test_trainer_lr_schedule_is_multistep_10_15():
    """configure_optimizers (BEVFusion_nuscenes_base_exp.py:436-441): AdamW(lr, wd 1e-7) + MultiStepLR [10, 15],
    stepped once per epoch."""
    from unidistill_amd import train
    dev = torch.device("cuda:0")
    tr = train.Trainer(train.DetectStep("lidar"), device=dev, lr=2e-4)
    assert tr.opt.defaults["weight_decay"] == 1e-7
    lrs = [tr.opt.param_groups[0]["lr"]] + [tr.epoch_end() for _ in range(16)]
    assert lrs[0] == lrs[9] == 2e-4
    assert abs(lrs[10] - 2e-5) < 1e-12 and abs(lrs[14] - 2e-5) < 1e-12 and abs(lrs[15] - 2e-6) < 1e-13
    state = tr.state_dict()
    tr2 = train.Trainer(train.DetectStep("lidar", model=tr.module.model), device=dev, lr=2e-4)
    tr2.load_state_dict(state)
    assert tr2.epoch == 16 and abs(tr2.opt.param_groups[0]["lr"] - 2e-6) < 1e-13
    batch = train.synthetic_batch(dev, batch_size=1, with_imgs=False)
    assert torch.isfinite(tr2.step(batch)["loss"])               # fused AdamW takes the scheduled lr


def test_cfg5_camera_student_fusion_teacher_bf16_10_sweeps():
    """BASELINE configs[4]: fusion teacher + camera student, bf16 mixed precision, 10-sweep LiDAR clouds."""
    step, out = _distill_step_checks("camera_exp_distill_fusion", batch_size=1, sweeps=10, ac=torch.bfloat16)
    assert step.teacher_model.fusion_encoder is not None
    assert step.exp["clamp"] == 1e-3


def test_cfg1_camera_only_student_one_camera():
    """BASELINE configs[0]: camera-only student, 1 camera 256x704, batch 1, no teacher (detector step)."""
    from unidistill_amd import train
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    step = train.DetectStep("camera")
    tr = train.Trainer(step, device=dev, channels_last=True)
    batch = train.synthetic_batch(dev, batch_size=1, ncam=1, with_points=False)
    assert batch["imgs"].shape == (1, 1, 1, 3, 256, 704)
    before = [p.detach().clone() for p in tr.params]
    out = tr.step(batch)
    assert torch.isfinite(out["loss"])
    assert sum(int(not torch.equal(a, b)) for a, b in zip(before, tr.params)) >= 0.95 * len(before)
    assert _finite_grads(step.model.parameters()) == len(tr.params)


@pytest.mark.parametrize("autocast", [None, torch.bfloat16])
def test_the_forward_pass_sees_the_optimizer_updates(autocast):
    """Every cached re-layout of a weight (Winograd filters, tap-major / transposed copies, bf16 casts, folded BatchNorm) must
    follow the parameter: fused optimizers step parameters WITHOUT moving their version counters, so a version-keyed cache of a
    trainable weight silently trains against stale filters.  Check end to end: on a fixed batch the loss of the LiDAR detector
    falls over a few steps, and the student's BEV features for the same input change after every step."""
    from unidistill_amd import train
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    tr = train.Trainer(train.DetectStep("lidar"), device=dev, autocast_dtype=autocast, channels_last=True)
    batch = train.synthetic_batch(dev, batch_size=2, with_imgs=False)
    losses = [float(tr.step(batch)["loss"].detach()) for _ in range(8)]
    assert all(l == l for l in losses)
    assert min(losses[-3:]) < 0.9 * losses[0], losses
    model = tr.module.model if hasattr(tr.module, "model") else tr.module
    pts = [p for p in batch["points"]]

    def bev():
        model.eval()
        with torch.no_grad(), torch.autocast("cuda", dtype=autocast or torch.float32, enabled=autocast is not None):
            out = model.extract_bev(pts, None, None)
        model.train()
        return (out[0] if isinstance(out, (tuple, list)) else out).float().clone()
    f0 = bev()
    tr.step(batch)
    f1 = bev()
    assert not torch.equal(f0, f1)


def test_nchw_camera_model_runs_through_the_library_when_not_strict(lenient):
    """An NCHW (not channels-last) camera model is outside the hand-written kernels' layout: under UD_STRICT it raises (see
    test_no_library_gpu.py), without it the step runs on the library path and is finite."""
    from unidistill_amd import train
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    tr = train.Trainer(train.DetectStep("camera"), device=dev)
    out = tr.step(train.synthetic_batch(dev, batch_size=1, ncam=1, with_points=False))
    assert torch.isfinite(out["loss"])
