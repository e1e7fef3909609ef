"""GPU parity of the COMPOSED path against goldens executed by the reference's own code (make_goldens.py
gold_model_step / gold_fusion_encoder, shrunk configuration tests/golden/shrunk.py):

  a7  LSSFPN._forward_single_sweep  (lss_fpn.py:266-320): lifted tensor, bins, pooled BEV map, depth
  a11 FusionEncoder                 (BEVFusion_nuscenes_base_exp.py:107-135), fp32 and the bf16 MFMA path
  a20 BEVFusionCenterHead.forward   (centerhead_fusion_exp.py:134-171), train and return_feature modes
  a20 Exp.training_step             (camera_exp_distill_lidar.py:438-513): valid-box scan, label +1, corner
      scaling, the three distillation losses, weights 100 / 40 / 10 -- loss, loss terms and EVERY student
      gradient.
Tolerances (fp32 end to end, summation orders differ from torch's CPU kernels): values 1e-4 relative,
gradients 2e-3 of the tensor's max."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
import shrunk as S  # noqa: E402

pytestmark = pytest.mark.gpu


def _sd(g, prefix):
    return {k[len(prefix):]: torch.from_numpy(g[k]) for k in g.files if k.startswith(prefix)}


def _register():
    from unidistill_amd.layers import image
    image.BACKBONES["TinyBackbone"] = S.TinyBackbone
    image.NECKS["TinyNeck"] = S.TinyNeck


def _model(g, tag):
    from unidistill_amd.models import BEVFusionCenterHead
    _register()
    m = BEVFusionCenterHead(S.ours_model_cfg())
    res = m.load_state_dict(_sd(g, tag + "_sd/"), strict=True)      # the reference's keys load unchanged
    assert not res.missing_keys and not res.unexpected_keys
    return m.cuda()


def _batch(g):
    c = lambda k: torch.from_numpy(g[k]).cuda()
    B, ncam = g["sensor2ego"].shape[:2]
    mats = {"sensor2ego_mats": c("sensor2ego").unsqueeze(1), "intrin_mats": c("intrin").unsqueeze(1),
            "ida_mats": c("ida").unsqueeze(1), "bda_mat": c("bda"),
            "sensor2sensor_mats": torch.eye(4, device="cuda").repeat(B, 1, ncam, 1, 1)}
    return {"imgs": c("imgs"), "mats_dict": mats, "gt_boxes": c("gt_boxes"), "gt_labels": c("gt_labels")}


def _close(got, ref, rtol=1e-4, atol_frac=1e-5, what=""):
    got = got.detach().float().cpu().numpy() if torch.is_tensor(got) else np.asarray(got)
    np.testing.assert_allclose(got, ref, rtol=rtol, atol=atol_frac * max(float(np.abs(ref).max()), 1e-30), err_msg=what)


def test_lssfpn_forward_single_sweep_vs_reference(golden, hip_lib, lenient):
    from unidistill_amd.ops import lss
    g = golden("model_step")
    m = _model(g, "student").eval()
    b = _batch(g)
    enc = m.camera_encoder.backbone
    D, C = enc.depth_channels, enc.output_channels
    with torch.no_grad():
        # the pieces, at the reference's op boundary
        feats = enc.get_cam_feats(b["imgs"])[:, 0]
        depth_feature = enc.depth_net(feats.reshape(-1, *feats.shape[2:]))
        lifted = lss.lift(depth_feature.float(), D, C)                       # [B*ncam, D, fH, fW, C]
        _close(lifted.reshape(g["lss_lifted"].shape), g["lss_lifted"], what="lifted tensor handed to the pool op")
        c = lambda k: torch.from_numpy(g[k]).cuda()
        mats = lss.prepare_mats(c("sensor2ego"), c("intrin"), c("ida"), c("bda"), c("ida_inv"), c("intrin_inv"))
        fu, fv, fd = enc._frustum_axes()
        B, ncam = g["sensor2ego"].shape[:2]
        bins, _ = lss.geometry(mats, fu, fv, fd, B, ncam, enc._lo, enc._size)
        assert int((bins.cpu().numpy().reshape(g["lss_geom_xyz"].shape) != g["lss_geom_xyz"]).sum()) == 0
        # the module, end to end (fused lift+splat; inverses from torch.linalg.inv_ex on the device, the module default)
        bev, depth = enc._forward_single_sweep(0, b["imgs"], b["mats_dict"], is_return_depth=True)
        _close(depth, g["lss_depth"], what="depth distribution")
        _close(bev, g["lss_bev"], what="pooled BEV map")
        enc.materialise = True
        bev2 = enc._forward_single_sweep(0, b["imgs"], b["mats_dict"])
        enc.materialise = False
        _close(bev2, g["lss_bev"], what="pooled BEV map, reference op boundary (lift -> voxel_pooling)")
        assert torch.equal(bev2, bev), "fused lift+splat must equal lift -> voxel_pooling bit for bit"


def test_model_return_feature_mode_vs_reference(golden, hip_lib, lenient):
    g = golden("model_step")
    t = _model(g, "teacher").eval()
    t.det_head.dense_head.distill = True
    b = _batch(g)
    gt = torch.cat([b["gt_boxes"], (b["gt_labels"] + 1).unsqueeze(2)], 2)
    with torch.no_grad():
        feat, trunk, heads = t(None, b["imgs"], b["mats_dict"], gt, return_feature=True)
    _close(feat, g["teacher_feat"], what="teacher bev feature")
    _close(heads[0]["hm"], g["teacher_head0_hm"], rtol=2e-4, what="teacher task-0 heat map (raw logits)")
    assert trunk.shape[1] == 24 and len(heads) == len(S.TASKS)


@pytest.mark.parametrize("channels_last", [False, True])
def test_distill_training_step_vs_reference(golden, hip_lib, channels_last, lenient):
    """channels_last=True is the benchmark's layout: NHWC convolution weights, the head's packed tail output copied to planes for
    the fused detection-loss kernels (layers/center_head.py:_split), strided response distillation."""
    from unidistill_amd import train
    g = golden("model_step")
    student, teacher = _model(g, "student"), _model(g, "teacher")
    step = train.DistillStep("camera_exp_distill_lidar", student=student, teacher=teacher, geometry=S.GEOMETRY)
    step.overlap_teacher = False
    step.cuda().train()
    if channels_last:
        train.to_channels_last(step)
    assert not step.teacher_model.training and step.model.training
    out = step(_batch(g))
    out["loss"].backward()
    _close(out["loss"], g["loss"], what="total loss = rpn + 100 feat + 40 rel + 10 (cls + reg)")
    for k in ("loss_rpn", "loss_feature", "loss_bev_rel", "loss_resp_cls", "loss_resp_reg"):
        _close(out["tb"][k], g[k], what=k)
    # every gradient of the student, against the reference's autograd.  The packed head keeps its weights in
    # fused tensors; its state_dict hook re-expresses any per-parameter quantity under the reference's keys,
    # so the gradients are pushed through it (data <-> grad swapped for the duration of the call).
    params = [p for p in student.parameters()]
    saved = [p.data for p in params]
    for p in params:
        p.data = p.grad if p.grad is not None else torch.zeros_like(p.data)
    grads_by_ref_key = {k: v.detach().clone() for k, v in student.state_dict().items()}
    for p, d in zip(params, saved):
        p.data = d
    n, worst = 0, (0.0, "")
    gmax = max(float(np.abs(g[k]).max()) for k in g.files if k.startswith("grad/"))
    for key in [k for k in g.files if k.startswith("grad/")]:
        name = key[len("grad/"):]
        ref = g[key]
        assert name in grads_by_ref_key, name
        got = grads_by_ref_key[name].float().cpu().numpy()
        err = float(np.abs(got - ref).max())
        tol = 2e-3 * float(np.abs(ref).max()) + 1e-6 * gmax        # biases ahead of a train-mode BN: ~0 gradients
        worst = max(worst, (err / max(tol, 1e-30), name))
        assert err <= tol, (name, err, tol)
        n += 1
    assert n >= 90, n
    print(f"{n} gradient tensors checked; worst error / tolerance = {worst[0]:.3f} ({worst[1]})")
    # BatchNorm running statistics after the train-mode pass
    sd = student.state_dict()
    for k in g.files:
        if k.startswith("student_after/"):
            _close(sd[k[len("student_after/"):]], g[k], rtol=1e-4, what=k)


def test_distill_step_with_train_mode_teacher_vs_reference(golden, hip_lib, lenient):
    """under Lightning the registered teacher is flipped back to train mode by model.train(), so its
    BatchNorms use batch statistics (make_goldens.py: model_step_teacher_train, executed by the reference's training_step with
    the teacher in train mode and the per-step checkpoint reload).  DistillStep(teacher_train_mode=True) reproduces that:
    loss, the four distillation terms, every student gradient, and the teacher's buffers after two steps (the reload keeps
    them ONE momentum step from the checkpoint, not two)."""
    from unidistill_amd import train
    g, gt_ = golden("model_step"), golden("model_step_teacher_train")
    student, teacher = _model(g, "student"), _model(g, "teacher")
    step = train.DistillStep("camera_exp_distill_lidar", student=student, teacher=teacher, geometry=S.GEOMETRY,
                             teacher_train_mode=True)
    step.overlap_teacher = False
    step.cuda().train()
    assert step.teacher_model.training and step.model.training
    ckpt = {k: v.detach().clone() for k, v in teacher.state_dict().items()}
    ckpt_buffers = {n_: b_.detach().clone() for n_, b_ in teacher.named_buffers()}
    out = step(_batch(g))
    out["loss"].backward()
    _close(out["loss"], gt_["loss"], what="total loss, teacher in train mode")
    for k in ("loss_rpn", "loss_feature", "loss_bev_rel", "loss_resp_cls", "loss_resp_reg"):
        _close(out["tb"][k], gt_[k], what=k)
    assert abs(float(out["tb"]["loss_feature"]) - float(g["loss_feature"])) > 1e-3 * abs(float(g["loss_feature"]))
    params = [p for p in student.parameters()]
    saved = [p.data for p in params]
    for p in params:
        p.data = p.grad if p.grad is not None else torch.zeros_like(p.data)
    grads_by_ref_key = {k: v.detach().clone() for k, v in student.state_dict().items()}
    for p, d in zip(params, saved):
        p.data = d
    gmax = max(float(np.abs(gt_[k]).max()) for k in gt_.files if k.startswith("grad/"))
    n = 0
    for key in [k for k in gt_.files if k.startswith("grad/")]:
        ref = gt_[key]
        got = grads_by_ref_key[key[len("grad/"):]].float().cpu().numpy()
        tol = 2e-3 * float(np.abs(ref).max()) + 1e-6 * gmax
        assert float(np.abs(got - ref).max()) <= tol, key
        n += 1
    assert n >= 90, n
    # second step on the same batch: the teacher's running statistics are reset first, so they end ONE update from the
    # checkpoint again (what the reference's state holds after any step)
    student.zero_grad(set_to_none=True)
    step(_batch(g))
    sd = teacher.state_dict()
    for k in gt_.files:
        if k.startswith("teacher_after/"):
            name = k[len("teacher_after/"):]
            if name.endswith("num_batches_tracked"):
                assert int(sd[name]) == int(gt_[k]), name
            else:
                _close(sd[name], gt_[k], rtol=1e-4, what=k)
    # a checkpoint loaded into the teacher AFTER steps have run (the reference: self.teacher_model.load_state_dict(ckpt),
    # ..._distill_lidar.py:424 -- a resume, a late teacher load) must not be undone by the buffers remembered before it: with
    # shifted running statistics loaded, the next step's reset starts from THEM
    shifted = {k: (v + 0.25 if k.endswith("running_mean") else v.clone()) for k, v in ckpt.items()}
    teacher.load_state_dict(shifted)
    student.zero_grad(set_to_none=True)
    step(_batch(g))
    step(_batch(g))
    sd = teacher.state_dict()
    momentum = {n_ + ".running_mean": m_.momentum for n_, m_ in teacher.named_modules() if hasattr(m_, "running_mean")}
    momentum.update({n_ + ".bn_running_mean": m_.bn_momentum for n_, m_ in teacher.named_modules() if hasattr(m_, "bn_momentum")})
    checked = 0
    for k in gt_.files:
        name = k[len("teacher_after/"):]
        if k.startswith("teacher_after/") and name.endswith("running_mean") and name in momentum:
            # one momentum step from the SHIFTED buffers instead of from the original ones: + (1 - momentum) * 0.25
            _close(sd[name], gt_[k] + (1 - momentum[name]) * 0.25, rtol=1e-4, what=k + " after a late load")
            checked += 1
    assert checked >= 5, checked
    # ... and buffers written from outside without any load (a manual edit) are picked up as well
    with torch.no_grad():
        for n_, b_ in teacher.named_buffers():
            b_.copy_(ckpt_buffers[n_])
    step(_batch(g))
    sd = teacher.state_dict()
    for k in gt_.files:
        name = k[len("teacher_after/"):]
        if k.startswith("teacher_after/") and name.endswith("running_mean"):
            _close(sd[name], gt_[k], rtol=1e-4, atol_frac=1e-5, what=k + " after a manual edit")


def test_student_forward_outputs_vs_reference(golden, hip_lib, lenient):
    g = golden("model_step")
    m = _model(g, "student").train()
    b = _batch(g)
    gt = torch.cat([b["gt_boxes"], (b["gt_labels"] + 1).unsqueeze(2)], 2)
    ret, tb, feat, trunk, heads, extra = m(None, b["imgs"], b["mats_dict"], gt)
    assert extra == {}
    _close(ret["loss"], g["loss_rpn"], what="loss_rpn")
    _close(feat, g["student_feat"], what="student bev feature")
    _close(trunk.mean((0, 2, 3)), g["student_trunk_mean"], rtol=2e-4, what="trunk channel means")
    _close(trunk.std((0, 2, 3)), g["student_trunk_std"], rtol=2e-4, what="trunk channel stds")
    for t in range(len(S.TASKS)):
        for hn, v in heads[t].items():
            _close(v, g[f"student_head{t}_{hn}"], rtol=2e-4, atol_frac=2e-5, what=f"head {t}/{hn}")


def test_fusion_encoder_gpu_fp32_and_bf16(golden, hip_lib):
    from unidistill_amd.layers.bev import FusionEncoder
    g = golden("fusion_encoder")
    m = FusionEncoder(use_elementwise=False, input_channel=128, output_channel=64)
    res = m.load_state_dict(_sd(g, "sd/"), strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    m = m.cuda()
    c = lambda k: torch.from_numpy(g[k]).cuda()
    x1, x2 = c("x1").requires_grad_(True), c("x2").requires_grad_(True)
    m.eval()
    with torch.no_grad():
        _close(m(x1, x2), g["y_eval"], what="eval fp32")
        with torch.autocast("cuda", dtype=torch.bfloat16):       # fused conv+BN+ReLU MFMA kernel
            y16 = m(x1.detach().contiguous(memory_format=torch.channels_last),
                    x2.detach().contiguous(memory_format=torch.channels_last))
        assert float((y16.float().cpu() - torch.from_numpy(g["y_eval"])).abs().max()) <= 2e-2 * float(np.abs(g["y_eval"]).max())
    m.train()
    y = m(x1, x2)
    _close(y, g["y_train"], what="train fp32")
    y.backward(c("gy"))
    _close(x1.grad, g["g1"], rtol=1e-3, atol_frac=1e-4, what="dx1")
    _close(x2.grad, g["g2"], rtol=1e-3, atol_frac=1e-4, what="dx2")
    _close(m.reduce_conv[0].weight.grad, g["gw"], rtol=1e-3, atol_frac=1e-4, what="dW reduce_conv")
    _close(m.att[1].weight.grad, g["gatt"], rtol=1e-3, atol_frac=1e-4, what="dW attention")
    # bf16 mixed-precision training path (MFMA conv fwd / dgrad / wgrad, streaming BN)
    m.zero_grad()
    a1 = c("x1").contiguous(memory_format=torch.channels_last).requires_grad_(True)
    a2 = c("x2").contiguous(memory_format=torch.channels_last).requires_grad_(True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y = m(a1, a2)
    y.float().backward(c("gy"))
    assert float((y.float().cpu() - torch.from_numpy(g["y_train"])).abs().max()) <= 2e-2 * float(np.abs(g["y_train"]).max())
    for got, key in ((a1.grad, "g1"), (a2.grad, "g2"), (m.reduce_conv[0].weight.grad, "gw")):
        ref = torch.from_numpy(g[key])
        cos = torch.nn.functional.cosine_similarity(got.float().cpu().flatten(), ref.flatten(), dim=0)
        assert cos > 0.999, (key, float(cos))
    _close(FusionEncoder(use_elementwise=True)(c("x1"), c("x2")), g["y_sum"], what="elementwise variant")
