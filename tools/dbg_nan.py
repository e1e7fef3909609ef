import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "cvpr2023-unidistill_amd")]
import torch
from unidistill_amd import train
dev = torch.device("cuda:0")
torch.manual_seed(0)
step = train.DetectStep("lidar").to(dev).train()
batch = train.synthetic_batch(dev, batch_size=1, with_imgs=False)
m = step.model
gt = torch.cat([batch["gt_boxes"], (batch["gt_labels"] + 1).unsqueeze(2)], 2)
with torch.no_grad():
    bev = m.extract_bev([p for p in batch["points"]], None, None)
bev = bev.detach().requires_grad_(True)
torch.autograd.set_detect_anomaly(True)
trunk, _ = m.bev_encoder(bev)
ret = m.det_head(trunk, gt)
d = ret["multi_head_features"][2]["dim"]
print("dim stats", d.min().item(), d.max().item(), torch.isfinite(d).all().item())
loss, tb = m.det_head.dense_head.get_loss(ret)
try:
    loss.backward()
except Exception as e:
    print("ANOMALY:", str(e)[:1500])
