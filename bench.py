"""bench.py -- UniDistill hot path on MI355X: one JSON line per run (driver contract).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one pass of the hot path over one per-GPU batch of synthetic nuScenes-shaped input
(SURVEY.md 8d).  The path shards by sample (pure data parallel): every rank works on its own
batch, so `scaling` is "weak" and value = (samples all ranks processed) / max-over-ranks time.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "cvpr2023-unidistill_amd")
for _p in (ROOT, PKG):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=1, help="samples per GPU")
    ap.add_argument("--workload", default="bev_extract", choices=["bev_extract"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    return ap.parse_args()


class BevExtract:
    """Camera splat + LiDAR voxelisation leg of the hot path, fwd + bwd of the splat.

    per sample: voxelize+MeanVFE of a 10-sweep cloud (300 k pts, cap 120 k voxels), and
    bev_pool fwd + bwd over the 6-camera frustum (N = 473 088 points, C = 256, 180x180 BEV).
    """
    name = "bev_extract(6cam bev_pool fwd+bwd C=256 180x180 + 10-sweep voxelize+meanVFE)"

    def __init__(self, device, batch, rank):
        from unidistill_amd import synthetic as syn
        from unidistill_amd.ops import bev_pool as bp, voxelize as vx
        self.bp, self.vx, self.syn = bp, vx, syn
        g = syn.rng(rank=rank)
        self.B = batch
        s2e, intr, ida, bda = syn.camera_rig(g, batch, 6)
        self.geom, _ = syn.frustum_bins_torch(s2e, intr, ida, bda, device)
        self.N = self.geom.shape[1]
        self.C, self.nx, self.ny = 256, 180, 180
        self.feat = torch.randn(batch, self.N, self.C, device=device)
        self.out = torch.empty(batch, self.ny, self.nx, self.C, device=device)
        self.pos = torch.empty(batch, self.N, 3, dtype=torch.int32, device=device)
        self.gout = torch.randn(batch, self.ny, self.nx, self.C, device=device).permute(0, 3, 1, 2)
        clouds = [syn.lidar_cloud(g, 30000, 10) for _ in range(batch)]
        self.points = torch.from_numpy(syn.pad_clouds(clouds)).to(device)
        self.dominant = "bev_pool.k_pool"
        # algorithmic bytes of one k_pool launch = the bev_pool fwd op's bytes
        self.alg_bytes = batch * self.N * (12 + self.C * 4 + 12) + batch * self.ny * self.nx * self.C * 4

    def step(self):
        bp = self.bp
        B, N, C = self.B, self.N, self.C
        self.vx.voxelize_batch(self.points, self.syn.VOXEL_SIZE, self.syn.POINT_CLOUD_RANGE, 10,
                               120000, want_voxels=False, want_mean=True)
        bp._pool_fwd(self.geom, self.feat, self.out, self.pos, B, N, C, self.nx, self.ny, 1,
                     bp.POOL_OVERWRITE)
        bp._pool_bwd(self.gout, self.pos, B, N, C, self.nx, self.ny)

    def cpu_baseline(self):
        """Oracle (scalar C, 1 thread) on a bounded sample: 1 camera of the splat fwd+bwd +
        a single-sweep voxelize; scaled to samples/s of the full step by work ratio."""
        import oracle
        syn = self.syn
        g = syn.rng(99)
        s2e, intr, ida, bda = syn.camera_rig(g, 1, 1)
        geom, _ = syn.frustum_bins_torch(s2e, intr, ida, bda, "cpu")
        geom = geom.numpy()
        n1 = geom.shape[1]
        feat = np.random.default_rng(0).standard_normal((1, n1, self.C)).astype(np.float32)
        gout = np.random.default_rng(1).standard_normal((1, self.C, self.ny, self.nx)).astype(np.float32)
        pts = syn.lidar_cloud(g, 30000, 1)
        t0 = time.perf_counter()
        reps = 0
        while time.perf_counter() - t0 < 10.0:
            _, pos = oracle.bev_pool_fwd(geom, feat, self.nx, self.ny, 1)
            oracle.bev_pool_bwd(gout, pos)
            oracle.voxelize(pts, syn.VOXEL_SIZE, syn.POINT_CLOUD_RANGE, 10, 120000, with_voxels=False)
            reps += 1
        dt = (time.perf_counter() - t0) / reps
        # full step = 6 cameras and 10 sweeps: 6x / 10x the sampled work
        est = 6.0 * dt
        return {"value": 1.0 / est, "unit": "samples/s", "cores": 1, "kind": "port",
                "sample": "oracle C, 1 thread: 1 of 6 cameras of bev_pool fwd+bwd (78848 pts x 256 ch) + "
                          "1-sweep (30k pt) voxelize+mean, time x6 extrapolated to the 6-cam step"}


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback for the product path)")
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    if world > 1:
        dist.init_process_group("nccl", device_id=device)
    from unidistill_amd import _lib
    _lib.load()
    wl = BevExtract(device, args.batch, rank)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        wl.step()
    barrier()
    _lib.prof_enable(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        wl.step()
    barrier()
    dt = time.perf_counter() - t0
    _lib.prof_enable(False)
    if world > 1:
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    k_ms, k_calls = _lib.prof_read(wl.dominant)
    if rank == 0:
        samples = args.batch * world * args.steps
        avg_us = (k_ms / max(k_calls, 1)) * 1e3
        achieved = wl.alg_bytes / (avg_us * 1e-6) / 1e9 if k_calls else None
        line = {
            "metric": "distill-train samples/sec (nuScenes frame); bev_pool+voxelize HBM GB/s",
            "value": samples / dt, "unit": "samples/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": wl.name, "batch_per_gpu": args.batch, "parallelism": f"dp{world}"},
            "roofline": {"bound": "hbm", "kernel": wl.dominant, "achieved": achieved,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": (achieved / HBM_PEAK_GBS) if achieved else None,
                         "avg_kernel_us": avg_us, "launches": k_calls,
                         "algorithmic_bytes_per_launch": wl.alg_bytes, "traffic": None},
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = wl.cpu_baseline()
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
