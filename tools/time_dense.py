"""SparseConvTensor.dense() (HeightCompression, a4) in fp32: ud_sparse_to_dense / ud_dense_to_sparse at the encoder's
output shape [B, 128, 2, 180, 180] with a synthetic occupancy; algorithmic bytes per SURVEY 8d."""
import os as _os; _os.environ.setdefault("UD_RANDOM_INIT", "1")   # synthetic weights (tools never train for real)
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "cvpr2023-unidistill_amd")]
import torch
from unidistill_amd import _lib
from unidistill_amd.ops import spconv as sp
d = torch.device("cuda:0")
for B in (1, 4):
    C, Dz, Hy, Wx = 128, 2, 180, 180
    torch.manual_seed(0)
    occ = torch.rand(B, Dz, Hy, Wx, device=d) < 0.35
    coords = occ.nonzero().int().contiguous()
    M = coords.shape[0]
    feat = torch.randn(M, C, device=d, requires_grad=True)
    x = sp.SparseConvTensor(feat, coords, (Dz, Hy, Wx), B)
    def fwd(): return x.dense()
    y = fwd(); g = torch.randn_like(y)
    def t(fn, n=20):
        for _ in range(3): fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n * 1e3
    tf = t(fwd)
    def bwd():
        feat.grad = None
        y.backward(g, retain_graph=True)
    tb = t(bwd)
    _lib.prof_enable(True); [fwd() for _ in range(10)]; torch.cuda.synchronize(); _lib.prof_enable(False)
    ms, n = _lib.prof_read("spconv.k_dense")
    alg = M * (C * 4 + 16) + B * C * Dz * Hy * Wx * 4
    print(f"B={B} M={M}: dense() op {tf:.1f} us ({alg/tf/1e6:.2f} TB/s algorithmic, {alg/tf/1e6/8*100:.0f}% of 8 TB/s); "
          f"k_dense kernel {ms/max(n,1)*1e3:.1f} us ({alg/(ms/max(n,1)*1e3)/1e6:.2f} TB/s); backward op {tb:.1f} us")
