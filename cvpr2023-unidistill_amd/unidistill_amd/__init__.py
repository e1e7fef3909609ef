"""unidistill_amd -- MI355X-native hot path for UniDistill (BEV extraction + distillation).

Host side mirrors the reference's operator interface (same names / argument meaning) on top of
the C-ABI library ``libunidistill_hip.so`` (``include/unidistill_hip.h``).  PyTorch is used for
device memory, streams and ``torch.distributed`` only.  There is NO CPU fallback: every op raises
if the HIP library is missing or a tensor is not on a GPU.
"""
from . import _lib  # noqa: F401

__all__ = ["_lib"]
