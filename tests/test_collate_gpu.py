"""GPU: camera input side + collate.
  * collate_fn on the device vs the golden produced by the reference's own collate_fn
    (nuscenes_multimodal.py:418-495): ragged clouds / boxes zero-padded, an empty box list, stacked matrices;
  * ud_image_normalize vs the oracle restatement of mmcv.imnormalize (third party, parity unpinned) -- bit-exact
    (float32 subtract then multiply, no contraction), both output layouts."""
import numpy as np
import pytest
import torch

import oracle

pytestmark = pytest.mark.gpu


def test_collate_fn_matches_reference_golden(golden, hip_lib):
    from unidistill_amd.ops import input_prep as ip
    g = golden("collate")
    data = []
    for i in range(3):
        d = {k: g[f"in{i}_{k}"] for k in ("imgs", "points", "gt_boxes", "gt_labels")}
        d["mats_dict"] = {k: g[f"in{i}_{k}"] for k in ("sensor2ego_mats", "intrin_mats", "ida_mats",
                                                       "sensor2sensor_mats", "bda_mat")}
        d["img_metas"] = {"token": f"t{i}"}
        data.append(d)
    out = ip.collate_fn(data, device="cuda")
    for k in ("imgs", "points", "gt_boxes", "gt_labels"):
        assert out[k].dtype == torch.float32 and out[k].is_cuda
        np.testing.assert_array_equal(out[k].cpu().numpy(), g["out_" + k])
    for k in ("sensor2ego_mats", "intrin_mats", "ida_mats", "sensor2sensor_mats", "bda_mat"):
        np.testing.assert_array_equal(out["mats_dict"][k].cpu().numpy(), g["out_" + k])
    assert [m["token"] for m in out["img_metas"]] == ["t0", "t1", "t2"]
    # also when the samples already live on the device
    data_dev = [dict(d, points=torch.from_numpy(d["points"]).cuda(), gt_boxes=torch.from_numpy(d["gt_boxes"]).cuda())
                for d in data]
    out2 = ip.collate_fn(data_dev, device="cuda")
    assert torch.equal(out2["points"], out["points"]) and torch.equal(out2["gt_boxes"], out["gt_boxes"])


@pytest.mark.parametrize("to_rgb", [True, False])
def test_image_normalize_bit_exact_vs_oracle(hip_lib, to_rgb):
    from unidistill_amd.ops import input_prep as ip
    rng = np.random.default_rng(5)
    imgs = rng.integers(0, 256, (2, 1, 3, 37, 53, 3), dtype=np.uint8)          # [B, sweeps, cams, H, W, 3]
    ref = oracle.image_normalize(imgs, ip.IMG_MEAN, ip.IMG_STD, to_rgb)         # HWC
    ref_chw = np.moveaxis(ref, -1, -3)
    x = torch.from_numpy(imgs).cuda()
    y = ip.image_normalize(x, to_rgb=to_rgb)
    assert y.shape == (2, 1, 3, 3, 37, 53) and y.dtype == torch.float32
    np.testing.assert_array_equal(y.cpu().numpy().view(np.int32), ref_chw.view(np.int32))
    ycl = ip.image_normalize(x, to_rgb=to_rgb, channels_last=True)
    assert ycl.shape == y.shape and torch.equal(ycl, y)
    assert ycl.reshape(-1, 3, 37, 53).is_contiguous(memory_format=torch.channels_last)
    # through collate_fn: uint8 samples are normalised on the device
    data = [{"imgs_u8": imgs[b], "gt_boxes": np.zeros((2, 9), np.float32), "gt_labels": np.zeros(2)} for b in range(2)]
    out = ip.collate_fn(data)
    ref_default = np.moveaxis(oracle.image_normalize(imgs, ip.IMG_MEAN, ip.IMG_STD, True), -1, -3)
    np.testing.assert_array_equal(out["imgs"].cpu().numpy(), ref_default)
