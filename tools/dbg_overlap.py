import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "cvpr2023-unidistill_amd")]
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
    names, gs = [], []
    for n, p in step.model.named_parameters():
        if p.grad is not None:
            names.append(n); gs.append(p.grad.flatten().float())
    return float(out["loss"]), names, gs
cos = lambda a, b: float(torch.nn.functional.cosine_similarity(torch.cat(a), torch.cat(b), dim=0))
l0, names, g0 = run(False)
l1, _, g1 = run(False)
l2, _, g2 = run(True)
print("loss", l0, l1, l2)
print("cos(no,no) =", cos(g0, g1), " cos(no,overlap) =", cos(g0, g2))
worst = sorted(((float(torch.nn.functional.cosine_similarity(a, b, dim=0)), float(a.norm()), n) for a, b, n in zip(g0, g1, names)))[:5]
print("least similar params between two identical runs:", worst)
