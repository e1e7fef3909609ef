"""CPU: host-side logic of the mirrors that needs no kernel -- the deferred pretrained-checkpoint check of the image backbone,
the cache invalidation hook, the trainer's LR schedule."""
import os

import pytest
import torch


def test_resnet_without_checkpoint_builds_but_refuses_to_train(monkeypatch):
    """mmdet's ResNet starts from torchvision://resnet50 (centerhead_fusion_exp.py:24-31) with a frozen stem.  Without a local
    checkpoint: building works (a full trained checkpoint may be loaded next), training on the random frozen stem raises, a
    load_state_dict clears the condition."""
    from unidistill_amd.layers.image import ResNet
    monkeypatch.setenv("UD_RANDOM_INIT", "0")
    monkeypatch.delenv("UD_RESNET50_CKPT", raising=False)
    torch.manual_seed(0)
    cfg = dict(depth=50, frozen_stages=0, out_indices=[0, 1, 2, 3], norm_eval=False,
               init_cfg=dict(type="Pretrained", checkpoint="torchvision://resnet50-not-here"))
    net = ResNet(**cfg)
    net.init_weights()
    assert not net.pretrained_loaded
    x = torch.randn(2, 3, 64, 64)
    net.eval()
    with torch.no_grad():
        assert len(net(x)) == 4                              # inference on whatever weights are loaded: allowed
    net.train()
    with pytest.raises(FileNotFoundError, match="frozen_stages=0"):
        net(x)
    net.load_state_dict(ResNet(**cfg).state_dict())          # e.g. a trained UniDistill checkpoint
    assert len(net(x)) == 4


def test_invalidate_caches_drops_every_weight_relayout():
    from unidistill_amd.ops import invalidate_caches
    m = torch.nn.Sequential(torch.nn.Conv2d(4, 4, 3), torch.nn.BatchNorm2d(4))
    m[0].weight._ud_wino = ("key", torch.zeros(1))
    m[0].weight._ud_tap_t = ("key", torch.zeros(1))
    m[1].running_var._ud_bn_eval = ("key", torch.zeros(1))
    m[0]._ud_stem_pack = ("key", torch.zeros(1))
    assert invalidate_caches(m) == 4
    assert not hasattr(m[0].weight, "_ud_wino") and not hasattr(m[1].running_var, "_ud_bn_eval")
    assert not hasattr(m[0], "_ud_stem_pack") and invalidate_caches(m) == 0


def test_bench_refuses_world_size_mismatch():
    """--gpus 8 under a 1-rank environment must not print a 1-GPU number as an 8-GPU one (the check precedes any GPU use)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    res = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--steps", "1", "--warmup", "0"],
                         cwd=root, env=env, capture_output=True, text=True, timeout=300)
    assert res.returncode != 0 and "WORLD_SIZE=1" in res.stderr and '{"metric"' not in res.stdout
