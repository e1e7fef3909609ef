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


_GRAPH_SCRIPT = r'''
import sys
sys.path[:0] = [{root!r}, {pkg!r}]
import torch
from unidistill_amd import train
dev = torch.device("cuda:0")
batch = train.synthetic_batch(dev, batch_size=1, ncam=6)
torch.manual_seed(0)
mod = train.DistillStep("camera_exp_distill_lidar")
state0 = {{k: v.clone() for k, v in mod.state_dict().items()}}
g = train.GraphTrainer(mod, batch, device=dev, warmup=1)            # warm-up restores the state
g.g_prep.replay(); g._reduce_norm(); g.lidar_bev.copy_(g._teacher_sparse()); g.g_tdense.replay()
g.g_student.replay(); torch.cuda.synchronize()
graph_loss = g.out["loss"].item()
graph_grads = {{n: p.grad.clone() for n, p in mod.model.named_parameters()}}
losses = [g.step(batch)["loss"].item() for _ in range(4)]
assert all(l == l for l in losses) and losses[-1] < losses[0], losses
# eager reference from the same initial state (after the graph work: no eager op precedes a replay)
ref = train.DistillStep("camera_exp_distill_lidar")
ref.load_state_dict(state0)
ref = ref.to(dev).train()
out = ref(batch)
out["loss"].backward()
assert abs(graph_loss - out["loss"].item()) <= 1e-4 * abs(out["loss"].item()), (graph_loss, out["loss"].item())
ref_grads = {{n: p.grad for n, p in ref.model.named_parameters() if p.grad is not None}}
gmax = max(v.double().norm().item() for v in ref_grads.values())
for n, b in ref_grads.items():
    a, b = graph_grads[n].flatten().double(), b.flatten().double()
    if b.norm() > 1e-3 * gmax:      # biases before a train-mode BN have a zero gradient (pure noise)
        cos = (a @ b / (a.norm() * b.norm() + 1e-30)).item()
        assert cos > 0.995, (n, cos)
        assert abs(a.norm().item() / b.norm().item() - 1) < 0.05, n
print("GRAPH_OK")
'''


def test_graph_trainer_matches_eager_trainer():
    """hipGraph-captured student pass == eager pass (loss + gradients of the first step) and the
    captured trainer keeps training.  Runs in a child process: on ROCm 7.x a hipGraph replay can
    fault when other eager torch ops ran in the process after capture (known issue);
    the captured trainer is opt-in (bench.py --graph), so a fault is reported as xfail, a wrong
    RESULT fails the test."""
    import os, subprocess, sys, signal
    from conftest import ROOT, PKG
    code = _GRAPH_SCRIPT.format(root=ROOT, pkg=PKG)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    if r.returncode in (-signal.SIGABRT, -signal.SIGSEGV, 134, 139) and "Memory access fault" in (r.stderr + r.stdout):
        pytest.xfail("hipGraph replay faulted in the runtime (known ROCm issue; GraphTrainer is opt-in)")
    assert r.returncode == 0 and "GRAPH_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


def test_teacher_on_second_stream_gives_the_same_step(hip_lib):
    """The frozen teacher on its own HIP stream (own scratch buffers, explicit hand-over) computes the same
    step.  The library convolutions of the student use split-K atomics, so two runs of the SAME
    configuration already differ by ~0.5 % at random init; the two modes must agree within that spread."""
    import torch
    from unidistill_amd import train
    dev = torch.device("cuda:0")

    def run(overlap):
        torch.manual_seed(0)
        step = train.DistillStep("camera_exp_distill_lidar").to(dev).train()
        step.overlap_teacher = overlap
        batch = train.synthetic_batch(dev, 1)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            out = step(batch)
        out["loss"].backward()
        torch.cuda.synchronize()
        g = torch.cat([p.grad.flatten() for p in step.model.parameters() if p.grad is not None])
        tb = {k: float(v) for k, v in out["tb"].items() if k.startswith("loss_")}
        return float(out["loss"]), g, tb

    base, g0, tb0 = run(False)
    over, g1, tb1 = run(True)
    assert abs(over - base) <= 0.03 * abs(base), (base, over)
    for k in ("loss_feature", "loss_bev_rel", "loss_resp_cls", "loss_resp_reg"):   # teacher-dependent terms
        assert abs(tb1[k] - tb0[k]) <= 0.03 * abs(tb0[k]) + 1e-6, (k, tb0[k], tb1[k])
    # (gradient directions are not comparable: at random init two runs of the same mode already have a
    #  cosine of ~0.3 through the library's atomically accumulated weight gradients, tools/dbg_overlap.py)
    assert torch.isfinite(g1).all() and g1.shape == g0.shape
