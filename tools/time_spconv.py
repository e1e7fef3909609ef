"""Time the LiDAR sparse encoder layers on a synthetic cloud (per-layer, fwd)."""
import os as _os; _os.environ.setdefault("UD_RANDOM_INIT", "1")   # synthetic weights (tools never train for real)
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "cvpr2023-unidistill_amd")]
import torch
from unidistill_amd import synthetic as syn, config as C, _lib
from unidistill_amd.layers.lidar import LidarEncoder
dev = torch.device("cuda:0")
B = int(os.environ.get("B", 1)); sweeps = int(os.environ.get("SWEEPS", 1))
enc = LidarEncoder(C.LIDAR_ENCODER).to(dev).eval()
g = syn.rng()
pts = [torch.from_numpy(syn.lidar_cloud(g, 30000, sweeps)).to(dev) for _ in range(B)]
n = min(p.shape[0] for p in pts); pts = [p[:n] for p in pts]
import contextlib
ac = torch.autocast('cuda', dtype=torch.bfloat16) if os.environ.get('AC') == 'bf16' else contextlib.nullcontext()
with torch.no_grad(), ac:
    for _ in range(3): enc(pts)
    torch.cuda.synchronize()
    _lib.prof_enable(True)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): out = enc(pts)
    e1.record(); torch.cuda.synchronize()
    _lib.prof_enable(False)
ms, n = _lib.prof_read("spconv.k_conv")
print(f"AC={os.environ.get('AC')} B={B} sweeps={sweeps}: encoder fwd {e0.elapsed_time(e1)/10:.2f} ms; spconv.k_conv total {ms/10:.2f} ms/fwd over {n//10} convs")
