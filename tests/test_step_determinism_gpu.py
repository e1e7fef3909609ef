"""GPU: run-to-run agreement of the distillation step -- the root cause analysis behind it is in DESIGN.md
("gradient agreement") and reproducible with tools/grad_cosine.py.

Findings the tests pin:
  * every hand-written kernel is order-deterministic (no fp atomics -- the last one, the scatter of the distillation
    losses' backward, was replaced by a sorted scatter at the end of round 2): with the libraries' strided /
    transposed convolutions switched to their deterministic algorithms the WHOLE bf16 step -- loss and every
    gradient -- is bitwise reproducible, also with the frozen teacher on a second stream (a race detector);
  * without that switch MIOpen's stride-2 convolutions differ by ~3e-6 from run to run, and the randomly
    initialised 50-layer train-mode-BatchNorm network amplifies any perturbation layer by layer (forward
    deviation bf16 vs fp32: 0.4 % after the stem, 57 % after layer4 -- identical for torch's own autocast path),
    which is what the cosine of ~0.3 between two nominally identical runs was;
  * against the fp32 step our bf16 kernels are at least as close as torch's bf16 autocast (library) path."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _run(ac, overlap=False, hip=True, tame=None, wl="camera_exp_distill_lidar"):
    from unidistill_amd import train
    from unidistill_amd.layers import dense
    dev = torch.device("cuda:0")
    dense.Conv2d.hip_enabled = hip
    try:
        torch.manual_seed(0)
        step = train.DistillStep(wl).to(dev).train()
        if tame is not None:            # damp the residual branches: a better-conditioned network
            with torch.no_grad():
                for n, m in step.named_modules():
                    if n.endswith(".bn3"):
                        m.weight.mul_(tame)
        step.overlap_teacher = overlap
        train.to_channels_last(step)
        batch = train.synthetic_batch(dev, 1)
        if ac is not None:
            with torch.autocast("cuda", dtype=ac):
                out = step(batch)
        else:
            out = step(batch)
        out["loss"].backward()
        torch.cuda.synchronize()
        grads = {n: p.grad.detach().float().flatten().clone() for n, p in step.model.named_parameters()
                 if p.grad is not None}
        return float(out["loss"]), grads
    finally:
        dense.Conv2d.hip_enabled = True


@pytest.fixture
def deterministic_libraries():
    prev = torch.backends.cudnn.deterministic
    torch.backends.cudnn.deterministic = True
    torch.use_deterministic_algorithms(True, warn_only=True)
    yield
    torch.use_deterministic_algorithms(False)
    torch.backends.cudnn.deterministic = prev


def test_bf16_step_is_bitwise_reproducible_with_deterministic_library_convs(hip_lib, deterministic_libraries):
    l0, g0 = _run(torch.bfloat16)
    l1, g1 = _run(torch.bfloat16)
    assert l0 == l1, (l0, l1)
    bad = [n for n in g0 if not torch.equal(g0[n], g1[n])]
    detail = [(n, float((g0[n] - g1[n]).abs().max()), float(g0[n].abs().max())) for n in bad]
    assert not bad, (len(bad), len(g0), detail[:3], detail[-6:])
    # the frozen teacher on its own stream (own scratch buffers, explicit hand-over): same bits
    l2, g2 = _run(torch.bfloat16, overlap=True)
    assert l2 == l0, (l0, l2)
    bad = [n for n in g0 if not torch.equal(g0[n], g2[n])]
    assert not bad, bad[:5]


def test_bf16_step_as_close_to_fp32_as_the_library_bf16_path(hip_lib, deterministic_libraries):
    """Tamed network (residual-branch gammas x0.2) so that the comparison measures kernels, not the chaotic
    amplification of rounding noise; large-norm gradients only."""
    cos = lambda a, b: float(torch.nn.functional.cosine_similarity(a.double(), b.double(), dim=0))
    _, ours = _run(torch.bfloat16, tame=0.2)
    _, lib = _run(torch.bfloat16, tame=0.2, hip=False)
    _, ref = _run(None, tame=0.2)
    gmax = max(float(v.norm()) for v in ref.values())
    big = [n for n, v in ref.items() if float(v.norm()) > 1e-3 * gmax]
    assert len(big) > 150
    c_ours = {n: cos(ours[n], ref[n]) for n in big}
    c_lib = {n: cos(lib[n], ref[n]) for n in big}
    cat = lambda g: torch.cat([g[n] for n in big])
    whole_ours, whole_lib = cos(cat(ours), cat(ref)), cos(cat(lib), cat(ref))
    print(f"whole-gradient cosine vs fp32: ours {whole_ours:.4f}, library bf16 {whole_lib:.4f}")
    assert whole_ours >= whole_lib - 0.03
    worse = [n for n in big if c_ours[n] < c_lib[n] - 0.15]
    assert len(worse) <= len(big) // 20, worse[:8]
    # one layer from the loss the bf16 gradient is essentially the fp32 one
    head = [n for n in big if "det_head" in n]
    assert head and min(c_ours[n] for n in head) > 0.9, {n: c_ours[n] for n in head}
    assert all(c_ours[n] >= c_lib[n] - 0.05 for n in head), {n: (c_ours[n], c_lib[n]) for n in head}


def test_bf16_step_is_bitwise_reproducible_by_construction(hip_lib):
    """Round 2 moved the strided / transposed convolutions of the bf16 step onto the hand-written kernels (the only
    library convolution left is the frozen, forward-only 7x7 stem), so two identical bf16 steps agree bit for bit
    WITHOUT asking the libraries for deterministic algorithms."""
    assert not torch.backends.cudnn.deterministic
    l0, g0 = _run(torch.bfloat16)
    l1, g1 = _run(torch.bfloat16)
    bad = [n for n in g0 if not torch.equal(g0[n], g1[n])]
    detail = [(n, float((g0[n] - g1[n]).abs().max()), float(g0[n].abs().max())) for n in bad]
    assert l0 == l1 and not bad, (l0, l1, len(bad), len(g0), detail[:3], detail[-6:])
