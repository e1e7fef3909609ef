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


def _run(ac, overlap=False, hip=True, tame=None, wl="camera_exp_distill_lidar", jitter=0.0):
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
        if jitter:                      # a perturbation far below fp32 kernel error budgets: the network's own sensitivity
            g = torch.Generator(device=dev).manual_seed(5)
            batch["imgs"] = batch["imgs"] + jitter * torch.randn(batch["imgs"].shape, device=dev, generator=g)
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


_FP32_STEP = r'''
import hashlib, json, os, sys, torch
sys.path[:0] = [{root!r}, {pkg!r}]
from unidistill_amd import train
from unidistill_amd.ops import wgrad_stream
dev = torch.device("cuda:0")
torch.manual_seed(0)
step = train.DistillStep("camera_exp_distill_lidar")
assert step.overlap_teacher and wgrad_stream.ENABLED
tr = train.Trainer(step, device=dev, channels_last=True)
batch = train.synthetic_batch(dev, 4)
sha = lambda t: hashlib.sha256(t.detach().contiguous().cpu().numpy().tobytes()).hexdigest()
out = tr.step(batch)
torch.cuda.synchronize()
rec = {{"loss0": float(out["loss"]).hex(),
       "grads": {{n: sha(p.grad) for n, p in step.model.named_parameters() if p.grad is not None}}}}
for _ in range(2):
    out = tr.step(batch)
torch.cuda.synchronize()
rec["loss2"] = float(out["loss"]).hex()
rec["params"] = sha(torch.cat([p.detach().flatten() for p in tr.params]))
rec["deferred"] = wgrad_stream.STATS["deferred"]
print("FP32_STEP " + json.dumps(rec))
'''


def test_fp32_headline_step_is_bitwise_reproducible_across_processes(hip_lib, tmp_path):
    """The benchmark's headline mode -- fp32, B = 4, weight gradients on their own stream, frozen teacher on a third -- in THREE
    fresh processes: the first step's loss and every gradient tensor, the third step's loss and all parameters after three
    optimizer steps agree bit for bit.  (Round 5 found a stale-fragment race in exactly this mode -- one launch in ~40 -- only
    because losses differed between processes; no in-process test could see it: this is that check as a test.)"""
    import json
    import os
    import subprocess
    import sys
    from conftest import PKG, ROOT
    path = tmp_path / "fp32_step.py"
    path.write_text(_FP32_STEP.format(root=ROOT, pkg=PKG))
    env = dict(os.environ, UD_RANDOM_INIT="1", UD_STRICT="1", UD_WGRAD_STREAM="1")
    recs = []
    for i in range(3):
        res = subprocess.run([sys.executable, str(path)], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
        assert res.returncode == 0, res.stdout[-1000:] + res.stderr[-3000:]
        line = [l for l in res.stdout.splitlines() if l.startswith("FP32_STEP ")][-1]
        recs.append(json.loads(line[len("FP32_STEP "):]))
    assert recs[0]["deferred"] > 100, recs[0]["deferred"]          # the weight-gradient stream really ran
    assert len(recs[0]["grads"]) > 150
    for r in recs[1:]:
        assert r["loss0"] == recs[0]["loss0"], (r["loss0"], recs[0]["loss0"])
        bad = [n for n in recs[0]["grads"] if r["grads"].get(n) != recs[0]["grads"][n]]
        assert not bad, (len(bad), bad[:8])
        assert r["loss2"] == recs[0]["loss2"] and r["params"] == recs[0]["params"]


def test_fp32_full_width_step_hand_kernels_vs_library(hip_lib):
    """The fp32 step at the REAL channel widths: every hand-written convolution / BatchNorm / head kernel (strict mode: no
    fall-through) against the same step with `dense.Conv2d.hip_enabled = False` (MIOpen / hipBLASLt / ATen), on the tamed network
    of the bf16 comparison above (residual gammas x 0.2: the comparison measures kernels, not the chaotic amplification of
    rounding noise by a randomly initialised 50-layer train-mode-BatchNorm network).  Loss 1e-4, whole-gradient cosine >= 0.999,
    and the detection head's gradients within 1e-3 of their max -- or within three times what a 1e-6 perturbation of the input
    images does to the LIBRARY path's own gradients, where that is larger: the head's gradients are sums of ReLU-masked terms
    over 32 400 pixels, the two paths round the trunk's output differently, a pre-activation within that distance of zero takes the
    other ReLU branch and moves a sum by a whole gradient value (measured: 2e-3 .. 1.4e-2 of a tensor's max between the paths,
    the same size as the perturbed library run's).  (The composed goldens run at shrunk widths, i.e. on library convolutions:
    this is the full-width step-level net under the hand kernels.)"""
    from unidistill_amd import _lib
    cos = lambda a, b: float(torch.nn.functional.cosine_similarity(a.double(), b.double(), dim=0))
    with _lib.strict(True):
        l_ours, ours = _run(None, tame=0.2, overlap=True)
    with _lib.strict(False):
        l_lib, lib = _run(None, tame=0.2, hip=False)
        _, lib_j = _run(None, tame=0.2, hip=False, jitter=1e-6)
    assert abs(l_ours - l_lib) <= 1e-4 * abs(l_lib), (l_ours, l_lib)
    assert set(ours) == set(lib)
    gmax = max(float(v.norm()) for v in lib.values())
    big = [n for n, v in lib.items() if float(v.norm()) > 1e-3 * gmax]
    assert len(big) > 150
    whole = cos(torch.cat([ours[n] for n in big]), torch.cat([lib[n] for n in big]))
    floor = cos(torch.cat([lib_j[n] for n in big]), torch.cat([lib[n] for n in big]))
    print(f"fp32 whole-gradient cosine, hand kernels vs library: {whole:.6f} (library vs library with 1e-6 input noise: {floor:.6f})")
    assert whole >= 0.999, whole
    head = [n for n in lib if "det_head" in n and float(lib[n].abs().max()) > 1e-6 * gmax]
    assert len(head) >= 8
    rel = lambda a, b: float((a - b).abs().max()) / float(b.abs().max())
    rows = sorted(((rel(ours[n], lib[n]), rel(lib_j[n], lib[n]), n) for n in head), reverse=True)
    for r in rows:
        print(f"  hand vs library {r[0]:.2e} of max; library, perturbed input {r[1]:.2e}  {r[2]}")
    # BatchNorm affine gradients (sums of 32 400 masked products that largely cancel) additionally differ by their summation order,
    # which a perturbed run of the SAME path does not show: 3e-3 for them (the kernels themselves are held to 1e-4 of the max on
    # well-conditioned data in tests/test_head_tail_gpu.py and tests/test_bn_act_gpu.py)
    affine = lambda n: n.endswith(("bn_weight", "bn_bias", ".1.weight", ".1.bias"))
    bad = [r for r in rows if r[0] > max(3e-3 if affine(r[2]) else 1e-3, 3 * r[1])]
    assert not bad, bad
