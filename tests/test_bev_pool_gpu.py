"""GPU parity: HIP bev_pool (through the C ABI) vs the CPU oracle and the reference goldens."""
import numpy as np
import pytest
import torch

import oracle

pytestmark = pytest.mark.gpu


def _dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _run_fwd(geom, feat, nx, ny, nz, overwrite):
    from unidistill_amd.ops import bev_pool as bp
    d = _dev()
    B, N, C = feat.shape
    g = torch.from_numpy(geom).to(d)
    f = torch.from_numpy(feat).to(d)
    if overwrite:
        out = torch.full((B, ny, nx, C), 7.0, device=d)       # garbage must be overwritten
        pos = torch.full((B, N, 3), 5, dtype=torch.int32, device=d)
    else:
        out = torch.zeros((B, ny, nx, C), device=d)
        pos = torch.full((B, N, 3), -1, dtype=torch.int32, device=d)
    bp._pool_fwd(g, f, out, pos, B, N, C, nx, ny, nz, bp.POOL_OVERWRITE if overwrite else bp.POOL_ACCUMULATE)
    torch.cuda.synchronize()
    return out.cpu().numpy(), pos.cpu().numpy()


def test_golden_fwd_bwd(golden):
    from unidistill_amd.ops.bev_pool import voxel_pooling
    g = golden("bev_pool")
    d = _dev()
    feat = torch.from_numpy(g["feat"]).to(d).requires_grad_(True)
    geom = torch.from_numpy(g["geom"]).to(d)
    out = voxel_pooling(geom, feat, (int(g["nx"]), int(g["ny"]), int(g["nz"])))
    assert out.shape == g["out_nchw"].shape
    np.testing.assert_allclose(out.detach().cpu().numpy(), g["out_nchw"], rtol=1e-6, atol=1e-6)
    out.backward(torch.from_numpy(g["gout_nchw"]).to(d))
    np.testing.assert_array_equal(feat.grad.cpu().numpy(), g["gfeat"])


def test_reference_wrapper_signature(golden):
    """voxel_pooling_forward_wrapper with caller-initialised outputs (lss_fpn.py:43-59)."""
    from unidistill_amd.ops.bev_pool import voxel_pooling_forward_wrapper
    g = golden("bev_pool")
    d = _dev()
    B, N, C = g["feat"].shape
    nx, ny, nz = int(g["nx"]), int(g["ny"]), int(g["nz"])
    out = torch.zeros(B, ny, nx, C, device=d)
    pos = torch.full((B, N, 3), -1, dtype=torch.int32, device=d)
    voxel_pooling_forward_wrapper(B, N, C, torch.tensor(nx), torch.tensor(ny), torch.tensor(nz),
                                  torch.from_numpy(g["geom"]).to(d), torch.from_numpy(g["feat"]).to(d),
                                  out, pos)
    np.testing.assert_array_equal(pos.cpu().numpy(), g["pos"])
    np.testing.assert_allclose(out.permute(0, 3, 1, 2).cpu().numpy(), g["out_nchw"], rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("B,N,C,nx,ny,nz", [
    (1, 1, 4, 1, 1, 1),            # single point
    (2, 1000, 256, 16, 12, 1),     # light + heavy cells mixed (avg 5/cell, hot spots below)
    (1, 5000, 80, 6, 5, 2),        # C not a multiple of 256, nz filter
    (1, 777, 7, 9, 9, 1),          # scalar (VEC=1) path
    (3, 4096, 256, 2, 2, 1),       # all cells heavy (>64 pts): LDS sort path
])
@pytest.mark.parametrize("overwrite", [False, True])
def test_random_vs_oracle_bitexact(B, N, C, nx, ny, nz, overwrite):
    rng = np.random.default_rng(B * 1000 + N + C)
    geom = np.stack([rng.integers(-1, nx + 1, (B, N)), rng.integers(-1, ny + 1, (B, N)),
                     rng.integers(-1, nz + 1, (B, N))], -1).astype(np.int32)
    feat = rng.standard_normal((B, N, C)).astype(np.float32)
    ref_out, ref_pos = oracle.bev_pool_fwd(geom, feat, nx, ny, nz)
    out, pos = _run_fwd(geom, feat, nx, ny, nz, overwrite)
    np.testing.assert_array_equal(pos, ref_pos)
    cnt = np.zeros((B, ny, nx), np.int64)
    kept = ref_pos[..., 0] >= 0
    np.add.at(cnt, (ref_pos[..., 0][kept], ref_pos[..., 1][kept], ref_pos[..., 2][kept]), 1)
    light = cnt <= 64
    # wave-per-cell path adds in ascending point order: bit-identical to the sequential oracle
    np.testing.assert_array_equal(out[light], ref_out[light])
    # heavy cells: 16 ordered partial sums -> fp32 reassociation only
    np.testing.assert_allclose(out[~light], ref_out[~light], rtol=2e-5, atol=2e-5)


def test_cell_counts_at_the_list_boundaries():
    """Cells holding exactly 0, 1, 63, 64 (last light count), 65 (first heavy: the reservation that crosses 64 lists the cell),
    255, 256 (a full row of the id table), 257 and 700 points (ranks past the row go through the overflow list), their points
    interleaved in the input so that every 256-point workgroup of k_bin sees many cells; two samples with different layouts."""
    rng = np.random.default_rng(21)
    counts = [0, 1, 63, 64, 65, 255, 256, 257, 700, 64, 65, 300]
    nx, ny, C = 4, 3, 256
    geoms, B = [], 2
    for b in range(B):
        cells = np.repeat(np.arange(len(counts)), np.roll(counts, b * 5))
        cells = np.concatenate([cells, np.full(500, -1)])          # + out-of-grid points
        rng.shuffle(cells)
        g = np.stack([np.where(cells >= 0, cells % nx, -1), np.where(cells >= 0, cells // nx, 0), np.zeros_like(cells)], -1)
        geoms.append(g)
    geom = np.stack(geoms).astype(np.int32)
    N = geom.shape[1]
    feat = rng.standard_normal((B, N, C)).astype(np.float32)
    ref_out, ref_pos = oracle.bev_pool_fwd(geom, feat, nx, ny, 1)
    for overwrite in (True, False):
        out, pos = _run_fwd(geom, feat, nx, ny, 1, overwrite)
        np.testing.assert_array_equal(pos, ref_pos)
        cnt = np.zeros((B, ny, nx), np.int64)
        kept = ref_pos[..., 0] >= 0
        np.add.at(cnt, (ref_pos[..., 0][kept], ref_pos[..., 1][kept], ref_pos[..., 2][kept]), 1)
        assert sorted(cnt[0].ravel().tolist()) == sorted(counts)
        light = cnt <= 64
        np.testing.assert_array_equal(out[light], ref_out[light])
        np.testing.assert_allclose(out[~light], ref_out[~light], rtol=2e-5, atol=2e-5)


def test_ultra_heavy_cell_scan_path():
    """> 8192 points in one cell exceeds the LDS sort and takes the index-range scan."""
    rng = np.random.default_rng(5)
    B, N, C = 2, 20000, 64
    geom = np.zeros((B, N, 3), np.int32)
    geom[1, ::3, 0] = 1                    # batch 1: two cells, one ~13.3k one ~6.7k
    geom[0, 10000:, 1] = 5                 # batch 0: half of the points out of range
    feat = rng.standard_normal((B, N, C)).astype(np.float32)
    ref_out, ref_pos = oracle.bev_pool_fwd(geom, feat, 2, 2, 1)
    out, pos = _run_fwd(geom, feat, 2, 2, 1, True)
    np.testing.assert_array_equal(pos, ref_pos)
    np.testing.assert_allclose(out, ref_out, rtol=1e-4, atol=1e-3)


def test_deterministic_and_empty_grid():
    rng = np.random.default_rng(9)
    B, N, C, nx, ny = 1, 30000, 256, 32, 32
    geom = np.stack([rng.integers(0, nx, (B, N)), rng.integers(0, ny, (B, N)),
                     np.zeros((B, N), np.int64)], -1).astype(np.int32)
    feat = rng.standard_normal((B, N, C)).astype(np.float32)
    a, _ = _run_fwd(geom, feat, nx, ny, 1, True)
    b, _ = _run_fwd(geom, feat, nx, ny, 1, True)
    np.testing.assert_array_equal(a, b)          # run-to-run bit reproducible
    geom[:] = -3                                  # nothing lands in the grid
    z, pos = _run_fwd(geom, feat, nx, ny, 1, True)
    assert (z == 0).all() and (pos == -1).all()


def test_bwd_layouts_vs_oracle():
    from unidistill_amd.ops import bev_pool as bp
    rng = np.random.default_rng(11)
    d = _dev()
    B, N, C, nx, ny = 2, 3000, 256, 20, 10
    geom = np.stack([rng.integers(-1, nx + 1, (B, N)), rng.integers(-1, ny + 1, (B, N)),
                     np.zeros((B, N), np.int64)], -1).astype(np.int32)
    feat = np.zeros((B, N, C), np.float32)
    _, pos = oracle.bev_pool_fwd(geom, feat, nx, ny, 1)
    gout = rng.standard_normal((B, C, ny, nx)).astype(np.float32)
    ref = oracle.bev_pool_bwd(gout, pos)
    tp = torch.from_numpy(pos).to(d)
    g_nchw = torch.from_numpy(gout).to(d)
    g_nhwc_view = g_nchw.permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)  # channels-last
    for g in (g_nchw, g_nhwc_view):
        got = bp._pool_bwd(g, tp, B, N, C, nx, ny)
        np.testing.assert_array_equal(got.cpu().numpy(), ref)


def test_full_size_properties():
    """BASELINE size (6 cams, N=473088, C=256, 180x180): linearity + mass conservation."""
    from unidistill_amd.ops import bev_pool as bp
    d = _dev()
    B, N, C, nx, ny = 1, 473088, 256, 180, 180
    gen = torch.Generator(device="cpu").manual_seed(3)
    geom = torch.stack([torch.randint(-20, nx + 20, (B, N), generator=gen),
                        torch.randint(-20, ny + 20, (B, N), generator=gen),
                        torch.zeros(B, N, dtype=torch.long)], -1).int().to(d)
    f1 = torch.randn(B, N, C, device=d)
    f2 = torch.randn(B, N, C, device=d)

    def run(f):
        out = torch.empty(B, ny, nx, C, device=d)
        pos = torch.empty(B, N, 3, dtype=torch.int32, device=d)
        bp._pool_fwd(geom, f, out, pos, B, N, C, nx, ny, 1, bp.POOL_OVERWRITE)
        return out, pos
    o1, pos = run(f1)
    o2, _ = run(f2)
    o12, _ = run(f1 + f2)
    assert torch.allclose(o12, o1 + o2, rtol=1e-4, atol=1e-4)            # linear in feat
    kept = pos[..., 0] >= 0
    mass_in = f1[kept].double().sum()
    assert abs(o1.double().sum().item() - mass_in.item()) < 1e-6 * f1[kept].abs().double().sum().item() + 1e-3
    ones, _ = run(torch.ones(B, N, C, device=d))                           # counts are exact ints
    assert ones[..., 0].sum().item() == kept.sum().item()
