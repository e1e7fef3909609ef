"""CPU, 2 gloo ranks: the N>1 path -- batched loss normalisers (reduce_mean_many) and DDP gradient
averaging of the detection head give the single-process answer."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

TASKS = [dict(num_class=1, class_names=["car"]), dict(num_class=2, class_names=["truck", "bus"])]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _head():
    from unidistill_amd.layers import center_head as ch
    pc, vs = [-16.0, -16.0, -5.0, 16.0, 16.0, 3.0], [0.25, 0.25, 0.2]
    names = ["car", "truck", "bus"]
    a = ch.FCOSAssigner(8, TASKS, 1, 0.1, 100, 2, {n: i + 1 for i, n in enumerate(names)}, [128, 128, 40],
                        pc[:2], vs[:2], 9, with_velocity=True)
    return ch.CenterHeadIouAware("nuscenes", TASKS, a, None, 8, 12, [128, 128, 40], pc, [1.0] * 8 + [0.2, 0.2],
                                 0.25, 5.0, 8, {"iou": [1, 2], "reg": [2, 2], "height": [1, 2], "dim": [3, 2],
                                                "rot": [2, 2], "vel": [2, 2]}, voxel_size_xy=vs[:2])


def _data(rank):
    g = torch.Generator().manual_seed(100 + rank)
    feat = torch.randn(1, 12, 16, 16, generator=g)
    n = 3 + 4 * rank                              # different number of boxes per rank
    gt = torch.zeros(1, 8, 10)
    gt[0, :n, 0:2] = (torch.rand(n, 2, generator=g) - 0.5) * 28
    gt[0, :n, 3:6] = torch.rand(n, 3, generator=g) * 2 + 0.5
    gt[0, :n, 6] = torch.rand(n, generator=g) * 6 - 3
    gt[0, :n, 9] = torch.randint(1, 4, (n,), generator=g).float()
    return feat, gt


def _loss(head, feat, gt):
    ret = head(feat, gt)
    for e in ret["box_encoding"].values():
        e[torch.isinf(e)] = 0
    return head.get_loss(ret)[0]


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from unidistill_amd import dist as ud
    torch.manual_seed(0)
    head = _head().train()
    # 1) batched normalisers == separate reduce_means
    vals = [torch.tensor(float(rank + 1)), torch.tensor(10.0 * (rank + 1))]
    many = ud.reduce_mean_many(vals)
    assert torch.allclose(torch.stack(many), torch.stack([ud.reduce_mean(v) for v in vals]))
    assert abs(many[0].item() - 1.5) < 1e-6 and abs(many[1].item() - 15.0) < 1e-6
    # 2) DDP: averaged gradients of the per-rank losses (which already use the global-mean normalisers)
    ddp = torch.nn.parallel.DistributedDataParallel(head)
    feat, gt = _data(rank)
    ret = ddp(feat, gt)
    for e in ret["box_encoding"].values():
        e[torch.isinf(e)] = 0
    loss = head.get_loss(ret)[0]
    loss.backward()
    if rank == 0:
        out["loss0"] = loss.item()
        out["grad"] = head.shared_conv[0].weight.grad.clone()
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_matches_single_process():
    port = _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    # single-process reference: emulate the global-mean normalisers by hand
    from unidistill_amd.layers import center_head as ch
    import unidistill_amd.layers.center_head as chm
    torch.manual_seed(0)
    head = _head().train()
    feats, gts = zip(*[_data(r) for r in range(2)])
    rets = []
    for f, g in zip(feats, gts):
        r = head(f, g)
        for e in r["box_encoding"].values():
            e[torch.isinf(e)] = 0
        rets.append(r)
    T = len(TASKS)
    local = [[r["heatmap"][t].eq(1).float().sum() for t in range(T)] + [r["mask"][t].float().sum() for t in range(T)]
             for r in rets]
    mean = [(local[0][i] + local[1][i]) / 2 for i in range(2 * T)]
    orig = chm.reduce_mean_many
    chm.reduce_mean_many = lambda xs: mean
    try:
        losses = [head.get_loss(r)[0] for r in rets]
    finally:
        chm.reduce_mean_many = orig
    assert abs(losses[0].item() - out["loss0"]) < 1e-4 * max(1.0, abs(out["loss0"]))
    ((losses[0] + losses[1]) / 2).backward()
    torch.testing.assert_close(head.shared_conv[0].weight.grad, out["grad"], rtol=1e-4, atol=1e-6)
