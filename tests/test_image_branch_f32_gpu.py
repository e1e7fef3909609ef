"""fp32 image branch as a WHOLE at its real size: ResNet-50 + SECONDFPN + depth net (the largest block of the headline step,
reference lss_fpn.py:143-171,242-250 / 277-288) on the hand-written kernels (Winograd 3x3, 1x1 / mapped convolutions, fused
BatchNorm statistics epilogues, streaming BN + ReLU, HIP stem) against the SAME module with ``Conv2d.hip_enabled = False``
(plain PyTorch / library ops), same weights, same input: forward 1e-4 of the output's max, every gradient 1e-3 of its max.
Errors of ~1e-6 per layer compound over 50 layers with training-mode BatchNorm in between; the kernel-level tests cannot
see that."""
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


def test_fp32_image_branch_hip_vs_library_forward_and_all_gradients(hip_lib):
    m = _branch(3)
    g = torch.Generator(device="cuda").manual_seed(5)
    imgs = torch.randn(1, 1, 6, 3, 256, 704, device="cuda", generator=g)
    proj = torch.randn(6, 368, 16, 44, device="cuda", generator=g)
    y_hip, g_hip, s_hip = _run(m, imgs, proj, True)
    y_lib, g_lib, s_lib = _run(m, imgs, proj, False)
    assert y_hip.shape == (6, 368, 16, 44)
    err = float((y_hip - y_lib).abs().max()) / float(y_lib.abs().max())
    assert err <= 1e-4, f"forward: {err:.3g} of max"
    assert set(g_hip) == set(g_lib) and len(g_hip) > 150
    worst = []
    for n, r in g_lib.items():
        scale = float(r.abs().max())
        e = float((g_hip[n].float() - r.float()).abs().max())
        worst.append((e / max(scale, 1e-30), n))
    worst.sort(reverse=True)
    assert worst[0][0] <= 1e-3, "gradients (err / max|ref|): " + ", ".join(f"{n} {e:.3g}" for e, n in worst[:8])
    for n, r in s_lib.items():                                        # BatchNorm statistics of every layer
        assert float((s_hip[n] - r).abs().max()) <= 1e-4 * float(r.abs().max()) + 1e-7, n
