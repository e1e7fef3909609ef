"""CPU, build container only: with unidistill_amd.shims installed the UNMODIFIED reference modules
import and construct on top of this library (skipped where /root/reference is absent, e.g. on the
GPU box; no kernels are launched)."""
import os
import sys
import types

import pytest
import torch

REF = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "unidistill")),
                                reason="reference tree not present")


def _install():
    from unidistill_amd import shims
    shims.install()
    # packages the reference imports for its data side only (not part of the hot path)
    for name in ("numba", "nuscenes", "pyquaternion", "skimage", "skimage.io", "cv2", "torchvision",
                 "torchvision.ops", "sklearn", "sklearn.datasets", "pytorch_lightning",
                 "pytorch_lightning.core"):
        if name not in sys.modules:
            m = types.ModuleType(name)
            m.__path__ = []
            sys.modules[name] = m
    sys.modules["numba"].jit = sys.modules["numba"].njit = lambda *a, **k: (a[0] if a and callable(a[0]) else (lambda f: f))
    sys.modules["torchvision.ops"].roi_align = None
    sys.modules["sklearn.datasets"].load_sample_images = None
    sys.modules["pyquaternion"].Quaternion = object
    if REF not in sys.path:
        sys.path.insert(0, REF)
    sys.dont_write_bytecode = True


def test_reference_modules_build_on_the_shims():
    _install()
    from unidistill.layers.blocks_3d.mmdet3d import lss_fpn
    from unidistill.layers.blocks_3d.det3d.spconv_backbone import VoxelResBackBone8x
    from unidistill.data.det3d.preprocess.voxelization import Voxelization
    from unidistill_amd.ops import spconv as sp, bev_pool
    from unidistill_amd import config as C
    assert lss_fpn.voxel_pooling_ext.voxel_pooling_forward_wrapper is bev_pool.voxel_pooling_forward_wrapper
    bb = VoxelResBackBone8x(5, __import__("numpy").array(C.GRID_SIZE))
    assert isinstance(bb.conv_input[0], sp.SubMConv3d) and isinstance(bb.conv2[0][0], sp.SparseConv3d)
    from unidistill_amd.layers.lidar import VoxelResBackBone8x as Mine
    mine = Mine(5, C.GRID_SIZE)
    assert {k: tuple(v.shape) for k, v in bb.state_dict().items()} == \
           {k: tuple(v.shape) for k, v in mine.state_dict().items()}
    vox = Voxelization(C.VOXEL_SIZE, C.POINT_CLOUD_RANGE, 10, (120000, 160000), 5, torch.device("cpu"))
    assert type(vox.voxel_generator).__name__ == "PointToVoxel"
    lss = lss_fpn.LSSFPN(**C.CAMERA_ENCODER)
    from unidistill_amd.layers.lss_fpn import LSSFPN as MineLSS
    ours = MineLSS(**C.CAMERA_ENCODER)
    assert {k: tuple(v.shape) for k, v in lss.state_dict().items()} == \
           {k: tuple(v.shape) for k, v in ours.state_dict().items()}
    torch.testing.assert_close(lss.frustum, ours.frustum, rtol=0, atol=0)
    torch.testing.assert_close(lss.voxel_coord, ours.voxel_coord, rtol=0, atol=0)
