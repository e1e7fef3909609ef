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


def _watch_one_step(tr, batch):
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


@pytest.mark.parametrize("autocast", [None, torch.bfloat16], ids=["fp32", "bf16"])
@pytest.mark.parametrize("workload,batch_size", [("camera_exp_distill_lidar", 2), ("lidar_exp_distill_fusion", 2),
                                                 ("camera_exp_distill_lidar", 1), ("camera_exp_distill_lidar", 4)])
def test_training_step_issues_no_library_conv_or_gemm(hip_lib, workload, batch_size, autocast):
    from unidistill_amd import _lib, train
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    with _lib.strict(True):          # and every fall-through branch of the package raises with its shape (UD_STRICT)
        tr = train.Trainer(train.DistillStep(workload), device=dev, autocast_dtype=autocast, channels_last=True)
        _watch_one_step(tr, train.synthetic_batch(dev, batch_size))


def test_cfg1_single_camera_detector_issues_no_library_op(hip_lib):
    """BASELINE configs[0] on the GPU: camera-only student, ONE camera, batch 1, no teacher (fp32)."""
    from unidistill_amd import _lib, train
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    with _lib.strict(True):
        tr = train.Trainer(train.DetectStep("camera"), device=dev, channels_last=True)
        _watch_one_step(tr, train.synthetic_batch(dev, 1, ncam=1, with_points=False))


def test_cfg5_ten_sweep_bf16_issues_no_library_op(hip_lib):
    """BASELINE configs[4]: fusion teacher + camera student, bf16 mixed precision, 10-sweep LiDAR."""
    from unidistill_amd import _lib, train
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    with _lib.strict(True):
        tr = train.Trainer(train.DistillStep("camera_exp_distill_fusion"), device=dev, autocast_dtype=torch.bfloat16,
                           channels_last=True)
        _watch_one_step(tr, train.synthetic_batch(dev, 2, sweeps=10))


def test_strict_mode_names_the_site_and_shape(hip_lib):
    """A layout no hand-written kernel takes (NCHW activations into a channels-last 3x3 convolution with 3 input channels) raises
    under UD_STRICT with the site and the shape instead of running MIOpen silently."""
    from unidistill_amd import _lib
    from unidistill_amd.layers.dense import Conv2d
    conv = Conv2d(3, 16, 5, padding=2).cuda()
    x = torch.randn(1, 3, 32, 32, device="cuda")
    with _lib.strict(True), pytest.raises(RuntimeError, match=r"UD_STRICT: layers.dense.Conv2d .*\(1, 3, 32, 32\)"):
        conv(x)
    with _lib.strict(False):
        assert conv(x).shape == (1, 16, 32, 32)
