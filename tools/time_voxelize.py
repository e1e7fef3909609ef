"""Quick GPU timing of voxelize(+mean) at the BASELINE cloud sizes."""
import os as _os; _os.environ.setdefault("UD_RANDOM_INIT", "1")   # synthetic weights (tools never train for real)
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "cvpr2023-unidistill_amd")]
import ctypes, numpy as np, torch
from unidistill_amd import synthetic as syn, _lib
from unidistill_amd.ops.voxelize import _f3

lib = _lib.load()
d = torch.device("cuda:0")
ALGO = int(os.environ.get("ALGO", "0"))      # 0: partition + LDS sort, 1: atomic hash, 2: three-launch hash with its memsets, 3: the same on its own clean workspace
def bench(B, sweeps, fused):
    g = syn.rng()
    pts = torch.from_numpy(syn.pad_clouds([syn.lidar_cloud(g, 30000, sweeps) for _ in range(B)])).to(d)
    B_, N, F = pts.shape
    P, maxM = 10, 120000
    cap = lib.ud_voxelize_capacity(B, N, maxM)
    ws = _lib.workspace(d, lib.ud_voxelize_workspace_bytes(B, N, P, maxM), "vox")
    vox = None if fused else torch.empty(cap, P, F, device=d)
    coords = torch.empty(cap, 4, dtype=torch.int32, device=d); num = torch.empty(cap, dtype=torch.int32, device=d)
    mean = torch.empty(cap, F, device=d); m = torch.empty(B + 2, dtype=torch.int32, device=d)
    vs, rg = _f3(syn.VOXEL_SIZE), _f3(syn.POINT_CLOUD_RANGE)
    st = _lib.stream_of(pts)
    if ALGO >= 2 and (B * N + 1023) // 1024 > 256:
        return
    def run(algo=ALGO):
        _lib.check(lib.ud_voxelize(_lib.ptr(pts), B, N, F, vs, rg, P, maxM, _lib.ptr(vox), _lib.ptr(coords),
                                   _lib.ptr(num), _lib.ptr(mean), _lib.ptr(m), _lib.ptr(ws), ws.numel(), algo, st), "vox")
    if ALGO == 3: run(2)      # leaves the workspace clean
    for _ in range(3): run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): run()
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 50 * 1e3
    M = int(m[B])
    alg = B * N * F * 4 + M * ((0 if fused else P * F * 4) + F * 4 + 16 + 4) if fused else B * N * 20 + M * 216
    print(f"B={B} sweeps={sweeps} N={N} M={M} fused={fused}: {t:.1f} us, algorithmic {alg/1e6:.2f} MB -> {alg/t/1e3:.1f} GB/s")
    _lib.prof_enable(True)
    for _ in range(10): run()
    torch.cuda.synchronize(); _lib.prof_enable(False)
    parts = []
    for k in (("k_partition", "k_bucket", "k_flags", "k_rows", "k_gather") if ALGO == 0 else ("k_insert", "k_first", "k_assign", "k_gather") if ALGO == 1 else ("k_insert2", "k_first_assign", "k_gather")):
        ms, n = _lib.prof_read("voxelize." + k)
        parts.append(f"{k} {ms / max(n, 1) * 1e3:.1f}")
    print("    per-kernel us (HIP events, incl. ~6 us dispatch each): " + ", ".join(parts))
for B, sw in ((1, 1), (4, 1), (1, 10), (4, 10)):
    for fused in (False, True):
        bench(B, sw, fused)
