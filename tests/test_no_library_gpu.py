"""GPU: the distillation training step reaches NO library convolution / GEMM, in fp32 (the headline) and in the bf16
mixed-precision mode: a TorchDispatchMode sees every ATen op of one whole step (forward, the autograd engine's backward,
optimizer) and the test fails with the op, its shapes and the package line that issued it.  A silent fallback to
MIOpen / hipBLASLt costs time (2.3 ms of the round-3 step were one) and breaks the step's bitwise reproducibility."""
import collections
import traceback

import pytest
import torch
from torch.utils._python_dispatch import TorchDispatchMode

pytestmark = pytest.mark.gpu

FORBIDDEN = ("aten.convolution", "aten._convolution", "aten.convolution_backward", "aten.miopen", "aten.cudnn",
             "aten.mm.", "aten.addmm", "aten.bmm", "aten.baddbmm", "aten.matmul", "aten.linear", "aten._conv",
             "aten.conv2d", "aten.conv_transpose", "aten.slow_conv", "aten.thnn_conv")
# 4 x 4 camera matrices (sensor2ego @ intrin^-1 ..., lss_fpn.py:212-231): the reference's own torch ops on [B, 6, 4, 4], kept
ALLOWED_MAX_ELEMS = 4 * 4 * 64


class _Watch(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.hits = collections.Counter()

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if name.startswith(FORBIDDEN):
            shapes = tuple(tuple(a.shape) for a in args if torch.is_tensor(a))
            if not all(torch.Size(s).numel() <= ALLOWED_MAX_ELEMS for s in shapes):
                site = "(autograd engine)"
                for fr in reversed(traceback.extract_stack()[:-1]):
                    if "unidistill_amd/" in fr.filename:
                        site = f"{fr.filename.split('unidistill_amd/')[-1]}:{fr.lineno} {fr.name}"
                        break
                self.hits[(name, shapes, site)] += 1
        return func(*args, **(kwargs or {}))


@pytest.mark.parametrize("autocast", [None, torch.bfloat16], ids=["fp32", "bf16"])
@pytest.mark.parametrize("workload", ["camera_exp_distill_lidar", "lidar_exp_distill_fusion"])
def test_training_step_issues_no_library_conv_or_gemm(hip_lib, workload, autocast):
    from unidistill_amd import train
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    tr = train.Trainer(train.DistillStep(workload), device=dev, autocast_dtype=autocast, channels_last=True)
    batch = train.synthetic_batch(dev, 2)
    for _ in range(2):
        tr.step(batch)
    torch.cuda.synchronize()
    w = _Watch()
    with w:
        out = tr.step(batch)
    torch.cuda.synchronize()
    assert torch.isfinite(out["loss"])
    report = "\n".join(f"{n:4d} x {name} {shapes} <- {site}" for (name, shapes, site), n in w.hits.most_common())
    assert not w.hits, "library convolution / GEMM ops in the step:\n" + report
