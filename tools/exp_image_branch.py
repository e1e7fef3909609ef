"""Diagnostic: fp32 image branch, HIP vs library vs an fp64 run of the same module (ground truth)."""
import copy, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "cvpr2023-unidistill_amd"), os.path.join(ROOT, "tests")]
os.environ.setdefault("UD_RANDOM_INIT", "1")
import torch
import test_image_branch_f32_gpu as T
from unidistill_amd.layers import dense, image
from unidistill_amd.ops import conv2d_f32 as c32

m = T._branch(3)
g = torch.Generator(device="cuda").manual_seed(5)
imgs = torch.randn(1, 1, 6, 3, 256, 704, device="cuda", generator=g)
proj = torch.randn(6, 368, 16, 44, device="cuda", generator=g)


def run64():
    m64 = copy.deepcopy(m).double()
    dense.Conv2d.hip_enabled = False
    image.ResNet.hip_stem = False
    try:
        feats = m64.get_cam_feats(imgs.double())[:, 0]
        depth = m64.depth_net(feats.reshape(6, *feats.shape[2:]))
        (depth * proj.double()).sum().backward()
        return depth.detach(), {n: p.grad.detach().clone() for n, p in m64.named_parameters() if p.grad is not None}
    finally:
        dense.Conv2d.hip_enabled = True
        image.ResNet.hip_stem = True


def cmp(tag, a, b):
    ya, ga = a[0], a[1]
    yb, gb = b[0], b[1]
    fe = float((ya.double() - yb.double()).abs().max() / yb.double().abs().max())
    worst = sorted(((float((ga[n].double() - r.double()).abs().max() / r.double().abs().max()), n) for n, r in gb.items()), reverse=True)
    print(f"{tag:34s} fwd {fe:.2e}  grads worst: " + ", ".join(f"{n.replace('img_backbone.', '')} {e:.2e}" for e, n in worst[:4]), flush=True)
    return worst


ref = run64()
hip = T._run(m, imgs, proj, True)
lib = T._run(m, imgs, proj, False)
lib2 = T._run(m, imgs, proj, False)
cmp("hip vs fp64", hip, ref)
cmp("lib vs fp64", lib, ref)
cmp("lib vs lib (2nd run)", lib2, lib)
cmp("hip vs lib", hip, lib)
w = cmp("hip vs fp64", hip, ref)
print("per-stage worst (hip vs fp64 | lib vs fp64):")
wl = dict((n, e) for e, n in cmp("lib vs fp64", lib, ref))
wh = dict((n, e) for e, n in w)
for key in ("layer1", "layer2", "layer3", "layer4", "img_neck", "depth_net"):
    hs = [wh[n] for n in wh if key in n]
    ls = [wl[n] for n in wl if key in n]
    print(f"  {key:10s} hip max {max(hs):.2e} median {sorted(hs)[len(hs)//2]:.2e} | lib max {max(ls):.2e} median {sorted(ls)[len(ls)//2]:.2e}")
for name, setter in (("no HIP BN", lambda v: setattr(dense, "_HIP_BN", not v)),
                     ("no winograd", lambda v: setattr(c32, "USE_WINOGRAD", not v)),
                     ("no hip wgrad", lambda v: setattr(c32, "USE_HIP_WGRAD", not v)),
                     ("no mapped", lambda v: setattr(dense, "_FP32_MAPPED", not v))):
    setter(True)
    try:
        cmp("hip [" + name + "] vs fp64", T._run(m, imgs, proj, True), ref)
    finally:
        setter(False)
