"""The cpu_baseline of bench.py (oracle/cpu_step.py: BASELINE configs[0] on the host = torch CPU ops + the CPU
oracle for geometry / lift / voxel pooling) computes the same step as the GPU product path: same weights,
same batch -> same loss and same gradients (fp32)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_cpu_restatement_of_cfg1_equals_the_gpu_step(hip_lib, lenient):
    import copy
    from oracle import cpu_step
    model, batch = cpu_step.build(seed=7)
    gpu = copy.deepcopy(model).cuda().train()
    loss_cpu = cpu_step.step(model, batch)
    dev = torch.device("cuda:0")
    mats = {"sensor2ego_mats": torch.from_numpy(batch["s2e"]).unsqueeze(1).to(dev),
            "intrin_mats": torch.from_numpy(batch["intr"]).unsqueeze(1).to(dev),
            "ida_mats": torch.from_numpy(batch["ida"]).unsqueeze(1).to(dev),
            "bda_mat": torch.from_numpy(batch["bda"]).to(dev)}
    ret, tb, *_ = gpu(None, batch["imgs"].to(dev), mats, batch["gt"].to(dev))
    ret["loss"].mean().backward()
    assert abs(float(ret["loss"]) - loss_cpu) <= 2e-4 * abs(loss_cpu), (float(ret["loss"]), loss_cpu)
    gc = {n: p.grad for n, p in model.named_parameters() if p.grad is not None}
    gg = {n: p.grad for n, p in gpu.named_parameters() if p.grad is not None}
    assert set(gc) == set(gg)
    gmax = max(float(v.norm()) for v in gc.values())
    for n in gc:
        a, b = gg[n].detach().float().cpu().flatten().double(), gc[n].flatten().double()
        if float(b.norm()) > 1e-3 * gmax:
            cos = float(a @ b / (a.norm() * b.norm()))
            assert cos > 0.99, (n, cos)     # random-init network: summation-order noise is amplified with depth
