"""LiDAR input side on the device (ud_points_transform) vs the reference golden and the numpy oracle."""
import numpy as np
import pytest
import torch

import oracle

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _info(g):
    return {"ego_to_global": g["key_ego_to_global"], "lidar_to_ego": g["key_lidar_to_ego"],
            "timestamp": int(g["timestamp"][0]),
            "sweep_lidar_infos": [{"sweep_lidar_to_ego": g[f"sweep{i}_lidar_to_ego"],
                                   "sweep_lidar_timestamp": int(g[f"sweep{i}_timestamp"][0])} for i in range(3)]}


def test_collect_lidar_sweeps_matches_reference(hip_lib, golden):
    from unidistill_amd.ops import input_prep as ip
    g = golden("input_prep")
    key = torch.from_numpy(g["key_points"]).to(DEV)
    sweeps = [torch.from_numpy(g[f"sweep{i}_points"]).to(DEV) for i in range(3)]      # one of them is empty
    got = ip.collect_lidar_sweeps(key, sweeps, _info(g)).cpu().numpy()
    np.testing.assert_array_equal(got, g["collected_points"])                         # bit-exact


def test_bev_affine_matches_reference(hip_lib, golden):
    from unidistill_amd.ops import input_prep as ip
    g = golden("input_prep")
    pts = torch.from_numpy(g["collected_points"]).to(DEV)
    boxes = torch.from_numpy(g["gt_boxes_in"]).to(DEV)
    for ci in range(4):
        a = g[f"bda{ci}_augs"]
        p2, b2, mat = ip.bev_affine(pts, boxes, a[0], a[1], a[2:5], bool(a[5]), bool(a[6]))
        np.testing.assert_array_equal(mat, g[f"bda{ci}_mat"])
        np.testing.assert_array_equal(p2.cpu().numpy(), g[f"bda{ci}_points"])
        np.testing.assert_array_equal(b2.cpu().numpy(), g[f"bda{ci}_boxes"])
    assert torch.equal(pts, torch.from_numpy(g["collected_points"]).to(DEV))          # inputs untouched


@pytest.mark.parametrize("D", [3, 4, 5])
def test_points_transform_large_batch_vs_oracle(hip_lib, D):
    """A 10-sweep x 4-sample batch (BASELINE size: ~300 k points per sample) in one launch: every segment
    its own matrix; ragged and empty segments; in-place operation."""
    from unidistill_amd.ops import input_prep as ip
    rng = np.random.default_rng(D)
    sizes = [int(v) for v in rng.integers(20000, 34000, size=44)]
    sizes[5], sizes[17] = 0, 1
    seg = np.cumsum([0] + sizes)
    pts = rng.normal(scale=[30.0, 30.0, 2.0] + [50.0] * (D - 3), size=(seg[-1], D)).astype(np.float32)
    mats = np.stack([oracle.sweep_to_key_matrix(*(np.linalg.qr(rng.normal(size=(4, 4)))[0] + np.eye(4) * 3
                                                  for _ in range(3))) for _ in sizes])
    last = rng.uniform(0, 0.5, size=len(sizes)).astype(np.float32)
    last[::3] = np.nan
    want = np.concatenate([oracle.points_transform(pts[a:b], m, None if np.isnan(l) else l)
                           for a, b, m, l in zip(seg, seg[1:], mats, last)])
    x = torch.from_numpy(pts).to(DEV)
    got = ip.points_transform(x, seg, mats, last)
    np.testing.assert_array_equal(got.cpu().numpy(), want)
    ip.points_transform(x, seg, mats, last, out=x)                                    # in place
    np.testing.assert_array_equal(x.cpu().numpy(), want)


def test_points_transform_rejects_bad_input(hip_lib):
    from unidistill_amd.ops import input_prep as ip
    x = torch.zeros(10, 5, device=DEV)
    with pytest.raises(ValueError):
        ip.points_transform(x, [0, 4], np.eye(4)[None])            # segments do not cover the rows
    with pytest.raises(ValueError):
        ip.points_transform(x.double(), [0, 10], np.eye(4)[None])
    with pytest.raises(RuntimeError):
        ip.points_transform(x.cpu(), [0, 10], np.eye(4)[None])     # no CPU fallback
    assert ip.points_transform(torch.zeros(0, 5, device=DEV), [0, 0], np.eye(4)[None]).shape == (0, 5)
