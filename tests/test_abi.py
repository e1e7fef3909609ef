"""CPU: the C-ABI library loads and exports every symbol include/unidistill_hip.h declares."""
import ctypes
import os
import re

from conftest import ROOT


def _declared():
    text = open(os.path.join(ROOT, "include", "unidistill_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ud_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_exported(hip_lib):
    names = _declared()
    assert len(names) >= 7
    for n in names:
        assert hasattr(hip_lib, n), f"{n} declared in unidistill_hip.h but not exported"


def test_binding_covers_header(hip_lib):
    from unidistill_amd import _lib
    assert set(_declared()) == set(_lib.exported_symbols())


def test_identity(hip_lib):
    assert hip_lib.ud_version().decode().startswith("unidistill_hip")
    assert hip_lib.ud_abi_version() >= 1
    assert hip_lib.ud_error_string(-2).decode().startswith("workspace")


def test_workspace_queries_are_pure(hip_lib):
    n = hip_lib.ud_bev_pool_workspace_bytes(1, 473088, 256, 180, 180, 1)
    assert n > 3 * 473088 * 4
    assert hip_lib.ud_bev_pool_workspace_bytes(0, 1, 1, 1, 1, 1) == 0


def test_no_cpu_fallback():
    """Product ops must refuse CPU tensors instead of silently computing elsewhere."""
    import pytest
    import torch
    from unidistill_amd.ops import bev_pool
    geom = torch.zeros(1, 4, 3, dtype=torch.int32)
    feat = torch.zeros(1, 4, 4)
    with pytest.raises(RuntimeError, match="GPU only"):
        bev_pool.voxel_pooling(geom, feat, (2, 2, 1))
