"""fp32 image branch as a WHOLE at its real size: ResNet-50 + SECONDFPN + depth net (the largest block of the headline step,
reference lss_fpn.py:143-171,242-250 / 277-288) on the hand-written kernels (Winograd 3x3, 1x1 / mapped convolutions, fused
BatchNorm statistics epilogues, streaming BN + ReLU, HIP stem) against the SAME module with ``Conv2d.hip_enabled = False``
(plain PyTorch / library ops) and with an fp64 run of it, same weights, same input.  Errors of ~1e-6 per layer compound over
50 layers with training-mode BatchNorm in between; the kernel-level tests cannot see that."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _branch(seed):
    from unidistill_amd import config as C, train
    from unidistill_amd.layers.lss_fpn import LSSFPN
    torch.manual_seed(seed)
    m = LSSFPN(**C.CAMERA_ENCODER).cuda()
    train.to_channels_last(m)
    with torch.no_grad():                       # BatchNorm affine parameters off their (1, 0) initial values
        for mod in m.modules():
            if isinstance(mod, torch.nn.BatchNorm2d):
                mod.weight.uniform_(0.6, 1.4)
                mod.bias.normal_(0, 0.2)
    return m.train()


def _run(m, imgs, proj, hip):
    from unidistill_amd.layers import dense, image
    dense.Conv2d.hip_enabled = hip
    image.ResNet.hip_stem = hip
    try:
        for mod in m.modules():
            if isinstance(mod, torch.nn.BatchNorm2d):
                mod.reset_running_stats()
        m.zero_grad(set_to_none=True)
        feats = m.get_cam_feats(imgs)[:, 0]
        B, ncam = feats.shape[:2]
        depth = m.depth_net(feats.reshape(B * ncam, *feats.shape[2:]))
        (depth.float() * proj).sum().backward()
        torch.cuda.synchronize()
        grads = {n: p.grad.detach().clone() for n, p in m.named_parameters() if p.grad is not None}
        stats = {n: b.detach().clone() for n, b in m.named_buffers() if n.endswith("running_var")}
        return depth.detach().float().clone(), grads, stats
    finally:
        dense.Conv2d.hip_enabled = True
        image.ResNet.hip_stem = True


def _run64(m, imgs, proj):
    """The same module in fp64 on plain PyTorch ops: the ground truth both fp32 paths are measured against."""
    import copy
    from unidistill_amd.layers import dense, image
    m64 = copy.deepcopy(m).double()
    dense.Conv2d.hip_enabled = False
    image.ResNet.hip_stem = False
    try:
        feats = m64.get_cam_feats(imgs.double())[:, 0]
        depth = m64.depth_net(feats.reshape(feats.shape[0] * feats.shape[1], *feats.shape[2:]))
        (depth * proj.double()).sum().backward()
        torch.cuda.synchronize()
        return depth.detach(), {n: p.grad.detach().clone() for n, p in m64.named_parameters() if p.grad is not None}
    finally:
        dense.Conv2d.hip_enabled = True
        image.ResNet.hip_stem = True


def _errs(grads, ref):
    return {n: float((grads[n].double() - r).abs().max()) / max(float(r.abs().max()), 1e-300) for n, r in ref.items()}


def test_fp32_image_branch_hip_vs_library_forward_and_all_gradients(hip_lib):
    """Forward: HIP == library to 1e-4 of the output's max (and both to the fp64 run).  Gradients: a RANDOMLY INITIALISED
    ResNet-50 under training-mode BatchNorm is ill-conditioned -- the library's own fp32 backward is 2-14 % (of each tensor's
    max) away from the fp64 gradient in layers 3-4 and 4-9 % away from ITS OWN second run (MIOpen's atomic split-K), measured
    on this box (tools/exp_image_branch.py) -- so a HIP-vs-library 1e-3 bound cannot hold for any fp32 implementation.  The
    bound that can: against the fp64 ground truth the hand-written path is as accurate as the library, tensor by tensor
    (err_hip <= 3 err_lib + 1e-3) and in the median (<= 1.5x), and where the problem is well conditioned (the depth net,
    the last layers before the output) it meets 1e-3 outright."""
    m = _branch(3)
    g = torch.Generator(device="cuda").manual_seed(5)
    imgs = torch.randn(1, 1, 6, 3, 256, 704, device="cuda", generator=g)
    proj = torch.randn(6, 368, 16, 44, device="cuda", generator=g)
    y64, g64 = _run64(m, imgs, proj)
    y_hip, g_hip, s_hip = _run(m, imgs, proj, True)
    y_lib, g_lib, s_lib = _run(m, imgs, proj, False)
    assert y_hip.shape == (6, 368, 16, 44)
    ymax = float(y64.abs().max())
    assert float((y_hip - y_lib).abs().max()) <= 1e-4 * ymax
    assert float((y_hip.double() - y64).abs().max()) <= 1e-4 * ymax
    assert set(g_hip) == set(g_lib) == set(g64) and len(g_hip) > 150
    e_hip, e_lib = _errs(g_hip, g64), _errs(g_lib, g64)
    bad = sorted(((e_hip[n], e_lib[n], n) for n in e_hip if e_hip[n] > 3 * e_lib[n] + 1e-3), reverse=True)
    assert not bad, "HIP gradient less accurate than the library's (err_hip, err_lib vs fp64): " + \
        ", ".join(f"{n} {a:.3g} {b:.3g}" for a, b, n in bad[:8])
    med = lambda d: sorted(d.values())[len(d) // 2]
    assert med(e_hip) <= 1.5 * med(e_lib) + 1e-4, (med(e_hip), med(e_lib))
    for n in e_hip:                                                    # well-conditioned end of the network: 1e-3 outright
        if n.startswith("depth_net"):
            assert e_hip[n] <= 1e-3, (n, e_hip[n])
    for n, r in s_lib.items():                                        # BatchNorm statistics of every layer
        assert float((s_hip[n] - r).abs().max()) <= 1e-4 * float(r.abs().max()) + 1e-7, n
