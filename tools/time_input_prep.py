"""ud_points_transform at the BASELINE batch (4 samples x 10 sweeps x ~30 k points, D=5) vs numpy on the host."""
import os as _os; _os.environ.setdefault("UD_RANDOM_INIT", "1")   # synthetic weights (tools never train for real)
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "cvpr2023-unidistill_amd")]
import numpy as np, torch
import oracle
from unidistill_amd import _lib
from unidistill_amd.ops import input_prep as ip
dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
sizes = [30000] * 44
seg = np.cumsum([0] + sizes)
pts = rng.normal(scale=[30, 30, 2, 50, 10], size=(seg[-1], 5)).astype(np.float32)
mats = np.stack([np.eye(4) + rng.normal(scale=0.1, size=(4, 4)) for _ in sizes])
last = rng.uniform(0, 0.5, len(sizes)).astype(np.float32)
x = torch.from_numpy(pts).to(dev)
for _ in range(3): ip.points_transform(x, seg, mats, last)
torch.cuda.synchronize()
_lib.prof_enable(True)
for _ in range(20): ip.points_transform(x, seg, mats, last)
torch.cuda.synchronize(); _lib.prof_enable(False)
ms, n = _lib.prof_read("input.k_points_transform")
us = ms / n * 1e3
mb = pts.nbytes * 2 / 1e6
print(f"{seg[-1]} points x 5: kernel {us:.1f} us = {mb / us * 1e3:.0f} GB/s ({mb:.0f} MB r+w)")
t0 = time.perf_counter()
for a, b, m, l in zip(seg, seg[1:], mats, last): oracle.points_transform(pts[a:b], m, l)
print(f"numpy (reference's code path, 1 core): {(time.perf_counter() - t0) * 1e3:.1f} ms")
