"""One plain fp32 1x1 shape through the persistent kernel for the SQ counter passes:
   bash tools/pmc_kernel.sh k_conv1x1p python tools/pmc_1x1p.py [P K N [use_ws]]"""
import ctypes as ct
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "cvpr2023-unidistill_amd")]
import torch
from unidistill_amd import _lib
P, K, N = [int(v) for v in sys.argv[1:4]] if len(sys.argv) > 3 else (16896, 256, 1024)
use_ws = len(sys.argv) > 4 and sys.argv[4] == "1"
d = torch.device("cuda:0")
lib = _lib.load()
x = torch.randn(P, K, device=d)
w = torch.randn(N, K, device=d) * 0.05
y = torch.empty(P, N, device=d)
ws = torch.empty(lib.ud_conv1x1p_f32_workspace_bytes(), dtype=torch.uint8, device=d)
for _ in range(13):
    _lib.check(lib.ud_conv1x1p_nhwc_f32(_lib.ptr(x), _lib.ptr(w), _lib.ptr(y), P, K, N, None, None, None, None, 0, None, 0, None,
                                        None, None, 0, 0, _lib.ptr(ws) if use_ws else None, ws.numel() if use_ws else 0,
                                        _lib.stream_of(x)), "ud_conv1x1p_nhwc_f32")
torch.cuda.synchronize()
