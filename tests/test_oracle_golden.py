"""CPU: pin the oracle (oracle/) against golden vectors captured from the reference python."""
import numpy as np

import oracle


def test_bev_pool_fwd_matches_reference_contract(golden):
    g = golden("bev_pool")
    out, pos = oracle.bev_pool_fwd(g["geom"], g["feat"], int(g["nx"]), int(g["ny"]), int(g["nz"]))
    np.testing.assert_array_equal(pos, g["pos"])
    # reference wrapper returns out.permute(0,3,1,2); index_add order == ascending point order
    np.testing.assert_allclose(out.transpose(0, 3, 1, 2), g["out_nchw"], rtol=1e-6, atol=1e-6)


def test_bev_pool_bwd_matches_reference(golden):
    g = golden("bev_pool")
    gfeat = oracle.bev_pool_bwd(g["gout_nchw"], g["pos"])
    np.testing.assert_array_equal(gfeat, g["gfeat"])


def test_mean_vfe_matches_reference(golden):
    g = golden("mean_vfe")
    np.testing.assert_allclose(oracle.mean_vfe(g["voxels"], g["num"]), g["out"], rtol=1e-6, atol=1e-7)


def test_voxelize_oracle_invariants():
    """spconv is not in the reference tree (parity unpinned): check the restated algorithm's own
    invariants -- every kept point's voxel matches its coordinates, first-appearance order, caps."""
    rng = np.random.default_rng(0)
    pts = rng.uniform(-60, 60, (2, 4000, 5)).astype(np.float32)
    pts[..., 2] = rng.uniform(-6, 4, (2, 4000))
    vs, rg = (1.5, 1.5, 2.0), (-54.0, -54.0, -5.0, 54.0, 54.0, 3.0)
    r = oracle.voxelize(pts, vs, rg, 3, 500)
    assert r["m"][0] == 500 and r["m"][1] == 500 and r["m"][2] == 1000
    assert r["num"].min() >= 1 and r["num"].max() == 3
    c = r["coords"]
    assert len(np.unique(c, axis=0)) == len(c)
    # slot 0 of each voxel is its first point: recompute its key and compare
    first = r["voxels"][:, 0, :3]
    key = np.floor((first - np.array(rg[:3], np.float32)) / np.array(vs, np.float32)).astype(np.int32)
    np.testing.assert_array_equal(key[:, ::-1], c[:, 1:])
    np.testing.assert_allclose(r["mean"], oracle.mean_vfe(r["voxels"], r["num"]), rtol=0, atol=0)


def test_lss_geometry_and_lift_oracle_vs_reference(golden):
    g = golden("lss_geometry")
    fr = g["frustum"]
    u, v, d = oracle.lss_frustum(tuple(g["final_dim"]), 16, tuple(g["d_bound"]))
    np.testing.assert_array_equal(u, fr[0, 0, :, 0])
    np.testing.assert_array_equal(v, fr[0, :, 0, 1])
    np.testing.assert_array_equal(d, fr[:, 0, 0, 2])
    # with the reference's own fp32 inverses: bit-identical coordinates, ZERO bin mismatches
    geom, bins = oracle.lss_geometry(g["sensor2ego"], g["intrin"], g["ida"], g["bda"], u, v, d,
                                     g["voxel_coord"], g["voxel_size"], g["ida_inv"], g["intrin_inv"])
    np.testing.assert_array_equal(geom.view(np.int32), g["geom"].view(np.int32))
    np.testing.assert_array_equal(bins, g["geom_xyz"])
    # with the correctly rounded inverse (the product default): the only difference is the last-bit
    # rounding of LAPACK's fp32 LU inverse; coordinates agree to a few ulp and the exact number of
    # bins that flip is printed and bounded
    geom2, bins2 = oracle.lss_geometry(g["sensor2ego"], g["intrin"], g["ida"], g["bda"], u, v, d,
                                       g["voxel_coord"], g["voxel_size"])
    np.testing.assert_allclose(geom2, g["geom"], rtol=2e-5, atol=2e-4)
    flips = int((bins2 != g["geom_xyz"]).any(-1).sum())
    print(f"exact-inverse variant: {flips} of {bins2[..., 0].size} points change bin vs the MKL-inverse golden")
    assert flips <= 1e-3 * bins2[..., 0].size and np.abs(bins2 - g["geom_xyz"]).max() <= 1
    l = golden("lss_lift")
    lifted, prob = oracle.lss_lift(l["depth_feature"], int(l["D"]), int(l["C"]))
    np.testing.assert_allclose(lifted.reshape(l["lifted"].shape), l["lifted"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(prob, l["depth"], rtol=1e-5, atol=1e-7)


def test_spconv_oracle_vs_dense_conv3d():
    """spconv is absent from the reference tree; pin the restated semantics against torch's dense
    conv3d evaluated on the site sets (submanifold, strided, reachable-set, dgrad, wgrad)."""
    import torch
    import torch.nn.functional as F
    rng = np.random.default_rng(0)
    shape = (2, 6, 9, 8)
    occ = rng.random(shape) < 0.3
    coords = np.argwhere(occ).astype(np.int32)
    rng.shuffle(coords)
    cin, cout = 5, 7
    feat = rng.standard_normal((len(coords), cin)).astype(np.float32)
    W = rng.standard_normal((cout, 27, cin)).astype(np.float32)
    bias = rng.standard_normal(cout).astype(np.float32)
    dense = torch.from_numpy(oracle.sparse_to_dense(feat, coords, shape)).requires_grad_(True)
    wt = torch.from_numpy(W).view(cout, 3, 3, 3, cin).permute(0, 4, 1, 2, 3).contiguous().requires_grad_(True)
    sel = lambda t, c: t[c[:, 0], :, c[:, 1], c[:, 2], c[:, 3]]
    nbr = oracle.spconv_subm_rulebook(coords, shape, (3, 3, 3))
    ref = F.conv3d(dense, wt, torch.from_numpy(bias), padding=1)
    np.testing.assert_allclose(oracle.spconv_conv(feat, nbr, W, bias), sel(ref, coords).detach().numpy(),
                               rtol=1e-5, atol=1e-5)
    gout = rng.standard_normal((len(coords), cout)).astype(np.float32)
    gd = torch.zeros_like(ref)
    gd[coords[:, 0], :, coords[:, 1], coords[:, 2], coords[:, 3]] = torch.from_numpy(gout)
    ref.backward(gd)
    np.testing.assert_allclose(oracle.spconv_conv(gout, nbr, W, mirror=True, transpose=True),
                               sel(dense.grad, coords).numpy(), rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(oracle.spconv_wgrad(feat, nbr, gout, cout),
                               wt.grad.permute(0, 2, 3, 4, 1).reshape(cout, 27, cin).numpy(), rtol=1e-4, atol=1e-4)
    oc, onbr, inbr, oshape = oracle.spconv_down(coords, shape, (3, 3, 3), (2, 2, 2), (1, 1, 1))
    ref2 = F.conv3d(dense.detach(), wt.detach(), None, stride=2, padding=1)
    np.testing.assert_allclose(oracle.spconv_conv(feat, onbr, W), sel(ref2, oc).numpy(), rtol=1e-5, atol=1e-5)
    reach = F.conv3d(torch.from_numpy(occ[:, None].astype(np.float32)), torch.ones(1, 1, 3, 3, 3),
                     stride=2, padding=1).numpy()[:, 0] > 0
    np.testing.assert_array_equal(np.argwhere(reach).astype(np.int32), oc)
    assert ((inbr >= 0).sum() == (onbr >= 0).sum())


def test_input_prep_oracle_matches_reference(golden):
    """CollectLidarSweeps / BevAffineTransformation restatements vs the reference's own outputs: bit-exact
    (float64 matrix products rounded once into the float32 cloud)."""
    g = golden("input_prep")
    sweeps = [g[f"sweep{i}_points"] for i in range(3)]
    got = oracle.collect_lidar_sweeps(g["key_points"], sweeps, g["key_lidar_to_ego"], g["key_ego_to_global"],
                                      g["timestamp"][0], [g[f"sweep{i}_lidar_to_ego"] for i in range(3)],
                                      [g[f"sweep{i}_timestamp"][0] for i in range(3)])
    np.testing.assert_array_equal(got, g["collected_points"])
    assert np.all(got[:1500, 4] == 0.0) and np.all(got[1500:, 4] > 0.04)     # key frame lag 0, sweeps ~50 ms apart
    for ci in range(4):
        a = g[f"bda{ci}_augs"]
        boxes, mat = oracle.bev_transform_boxes(g["gt_boxes_in"], a[0], a[1], a[2:5], bool(a[5]), bool(a[6]))
        np.testing.assert_array_equal(mat, g[f"bda{ci}_mat"])
        np.testing.assert_array_equal(boxes, g[f"bda{ci}_boxes"])
        np.testing.assert_array_equal(oracle.points_transform(g["collected_points"], mat), g[f"bda{ci}_points"])


def test_input_prep_host_matrices_match_reference(golden):
    """Host half of ops/input_prep.py (no GPU needed): the float64 matrices it hands to ud_points_transform
    are the reference's, bit for bit."""
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "cvpr2023-unidistill_amd"))
    from unidistill_amd.ops import input_prep as ip
    g = golden("input_prep")
    for ci in range(4):
        a = g[f"bda{ci}_augs"]
        np.testing.assert_array_equal(ip.bev_transform_matrix(float(a[0]), float(a[1]), a[2:5], bool(a[5]), bool(a[6])),
                                      g[f"bda{ci}_mat"])
    for i in range(3):
        m = ip.sweep_to_key_matrix(g["key_lidar_to_ego"], g["key_ego_to_global"], g[f"sweep{i}_lidar_to_ego"])
        np.testing.assert_array_equal(m, oracle.sweep_to_key_matrix(g["key_lidar_to_ego"], g["key_ego_to_global"],
                                                                    g[f"sweep{i}_lidar_to_ego"]))


def test_collate_oracle_matches_reference_golden(golden):
    g = golden("collate")
    for k in ("imgs", "points", "gt_boxes", "gt_labels"):
        np.testing.assert_array_equal(oracle.collate_fill([g[f"in{i}_{k}"] for i in range(3)]), g["out_" + k])


def test_openmp_build_of_the_oracle_equals_the_scalar_checker():
    """bench.py's cpu_baseline runs bev_pool forward / backward and the lift from the OpenMP build of oracle/ud_oracle.c
    (oracle.use_openmp): same bits as the scalar checker (rows are owned by threads, every cell still sums in point order); the
    C lift equals the numpy formulation."""
    import numpy as np
    import oracle
    rng = np.random.default_rng(3)
    B, N, C, nx, ny = 2, 5000, 16, 12, 10
    geom = np.stack([rng.integers(-2, nx + 2, (B, N)), rng.integers(-2, ny + 2, (B, N)), rng.integers(0, 2, (B, N))], -1).astype(np.int32)
    feat = rng.standard_normal((B, N, C)).astype(np.float32)
    out0, pos0 = oracle.bev_pool_fwd(geom, feat, nx, ny, 1)
    g = rng.standard_normal((B, C, ny, nx)).astype(np.float32)
    gf0 = oracle.bev_pool_bwd(g, pos0)
    x = rng.standard_normal((3, 7 + 5, 4, 6)).astype(np.float32)
    l0, p0 = oracle.lss_lift(x, 7, 5)
    prev = oracle.use_openmp(True)
    try:
        out1, pos1 = oracle.bev_pool_fwd(geom, feat, nx, ny, 1)
        gf1 = oracle.bev_pool_bwd(g, pos1)
        l1, p1 = oracle.lss_lift(x, 7, 5)
    finally:
        oracle.use_openmp(prev)
    assert np.array_equal(out0, out1) and np.array_equal(pos0, pos1) and np.array_equal(gf0, gf1)
    np.testing.assert_allclose(l1, l0, rtol=2e-6, atol=1e-7)
    np.testing.assert_allclose(p1, p0, rtol=2e-6, atol=1e-7)
