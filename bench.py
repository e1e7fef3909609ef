"""bench.py -- UniDistill distillation training on MI355X: one JSON line per run (driver contract).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one full distillation training step (student fwd + frozen-teacher fwd + detection and
distillation losses + backward + grad clip + AdamW) over one per-GPU batch of synthetic
nuScenes-shaped input.  The path is pure data parallel: every rank trains on its
own samples and gradients are averaged with RCCL, so `scaling` is "weak" and
value = samples of all ranks / max-over-ranks wall time.

After the timed region (never counted in `value`) two extra legs run on rank 0:
  * roofline: the reference-boundary bev_pool forward (BASELINE.json: "bev_pool+voxelize HBM GB/s")
    at the BASELINE shape, its dominant kernel timed with HIP events on its launch stream;
  * cpu_baseline: the CPU oracle (oracle/, scalar C) on a bounded sample of the same hot path.
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

WORKLOADS = {
    # BASELINE.json configs[2]: camera student + LiDAR teacher, 6-cam nuScenes shape
    "camera_exp_distill_lidar": dict(kind="distill", sweeps=1),
    "camera_exp_distill_fusion": dict(kind="distill", sweeps=10),   # configs[4] models
    "lidar_exp_distill_fusion": dict(kind="distill", sweeps=1),     # configs[3] models
    "lidar_exp_distill_camera": dict(kind="distill", sweeps=1),
    "lidar": dict(kind="detect", sweeps=1),                         # configs[1]
    "camera": dict(kind="detect", sweeps=1),
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=4,
                    help="samples per GPU (reference Exp default batch_size_per_device=4; BASELINE configs[3])")
    ap.add_argument("--workload", default="camera_exp_distill_lidar", choices=sorted(WORKLOADS))
    ap.add_argument("--autocast", default="bf16", choices=["none", "bf16"],
                    help="bf16 autocast for the dense (MIOpen) convs; the HIP ops always compute in fp32")
    ap.add_argument("--nchw", action="store_true", help="keep dense convs NCHW (default: channels-last)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    return ap.parse_args()


def roofline_leg(device, batch):
    """bev_pool forward (reference op boundary) at N = 6*112*16*44 points, C = 256, 180x180."""
    from unidistill_amd import _lib, synthetic as syn
    from unidistill_amd.ops import bev_pool as bp
    g = syn.rng(7)
    s2e, intr, ida, bda = syn.camera_rig(g, batch, 6)
    geom, _ = syn.frustum_bins_torch(s2e, intr, ida, bda, device)
    B, N = geom.shape[:2]
    C, nx, ny = 256, 180, 180
    feat = torch.randn(B, N, C, device=device)
    out = torch.empty(B, ny, nx, C, device=device)
    pos = torch.empty(B, N, 3, dtype=torch.int32, device=device)
    scrub = torch.empty(512 << 20, dtype=torch.uint8, device=device)   # > Infinity Cache (256 MiB)
    for _ in range(2):
        bp._pool_fwd(geom, feat, out, pos, B, N, C, nx, ny, 1, bp.POOL_OVERWRITE)
    torch.cuda.synchronize()
    _lib.prof_enable(True)
    reps = 10
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    op_ms = 0.0
    for _ in range(reps):
        scrub.zero_()                       # evict feat from the Infinity Cache: honest HBM reads
        e0.record()
        bp._pool_fwd(geom, feat, out, pos, B, N, C, nx, ny, 1, bp.POOL_OVERWRITE)
        e1.record()
        torch.cuda.synchronize()
        op_ms += e0.elapsed_time(e1)
    _lib.prof_enable(False)
    k_ms, k_calls = _lib.prof_read("bev_pool.k_pool")
    alg = B * N * (12 + C * 4 + 12) + B * ny * nx * C * 4          # SURVEY 8d bev_pool fwd row
    k_us = k_ms / max(k_calls, 1) * 1e3
    achieved = alg / (k_us * 1e-6) / 1e9
    return {"bound": "hbm", "kernel": "bev_pool.k_pool (ud_bev_pool_fwd, reference op boundary)",
            "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
            "avg_kernel_us": k_us, "launches": k_calls, "algorithmic_bytes_per_launch": alg,
            "op_avg_us": op_ms / reps * 1e3, "op_GBps": alg / (op_ms / reps * 1e-3) / 1e9,
            # HBM bytes per launch from the PMC passes of the same op at the same shape
            # (profiles/r01_pmc_k_pool.md: FETCH_SIZE 182401 KB x2 (gfx950) + WRITE_SIZE 32400 KB);
            # counters cannot be read from inside this process, so the committed figure is reported.
            "traffic": 406.8e6 if (C, nx, ny, N) == (256, 180, 180, 473088) else None,
            "note": "measured after the timed region; HIP events bracket each launch, so avg_kernel_us carries "
                    "the ~6 us dispatch latency that rocprofv3's kernel duration (profiles/) does not; the "
                    "training step itself uses the fused lift+splat (no [B,N,C] tensor), see DESIGN.md"}


MFMA_PEAK_TFLOPS = 2500.0   # MI355X_MICROARCH.md: dense bf16 MFMA peak (no sparsity)


def mfma_leg(trainer, batch, steps=3):
    """The dense trunk / head / image-branch 3x3 convolutions (k_conv3x3_bf16, the largest kernel family of
    the step) over a few extra training steps: algorithmic flops / HIP-event kernel time."""
    from unidistill_amd import _lib
    from unidistill_amd.ops import conv2d as c2
    c2.FLOP_COUNTER = [0]
    _lib.prof_read("conv2d.k_conv3x3", reset=True)
    _lib.prof_enable(True)
    for _ in range(steps):
        trainer.step(batch)
    torch.cuda.synchronize()
    _lib.prof_enable(False)
    ms, calls = _lib.prof_read("conv2d.k_conv3x3")
    flops, c2.FLOP_COUNTER = c2.FLOP_COUNTER[0], None
    if not calls or ms <= 0:
        return None
    achieved = flops / (ms * 1e-3) / 1e12
    return {"bound": "mfma", "kernel": "conv2d.k_conv3x3 (ud_conv3x3_nhwc_bf16: BEV trunk, head, ResNet 3x3 convs; fwd + dgrad)",
            "achieved": achieved, "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": achieved / MFMA_PEAK_TFLOPS,
            "launches": calls, "avg_kernel_us": ms / calls * 1e3, "algorithmic_flops_per_step": flops / steps,
            "kernel_ms_per_step": ms / steps, "traffic": None,
            "note": "measured over %d extra steps after the timed region (HIP events around every launch)" % steps}


def cpu_baseline_leg():
    """Oracle (scalar C, 1 thread) on a bounded sample: 1 of 6 cameras of lift + bev_pool fwd+bwd,
    a single-sweep voxelize+mean, a 2k-voxel slice of one 64->64 sparse conv, the 3 distill losses'
    mask; extrapolated by the work ratio to one distillation step of the BEV extraction path."""
    import oracle
    from unidistill_amd import synthetic as syn
    g = syn.rng(99)
    s2e, intr, ida, bda = syn.camera_rig(g, 1, 1)
    geom, _ = syn.frustum_bins_torch(s2e, intr, ida, bda, "cpu")
    geom = geom.numpy()
    n1 = geom.shape[1]
    rs = np.random.default_rng(0)
    feat = rs.standard_normal((1, n1, 256)).astype(np.float32)
    gout = rs.standard_normal((1, 256, 180, 180)).astype(np.float32)
    pts = syn.lidar_cloud(g, 30000, 1)
    t0 = time.perf_counter()
    reps = 0
    while time.perf_counter() - t0 < 12.0:
        _, pos = oracle.bev_pool_fwd(geom, feat, 180, 180, 1)
        oracle.bev_pool_bwd(gout, pos)
        oracle.voxelize(pts, syn.VOXEL_SIZE, syn.POINT_CLOUD_RANGE, 10, 120000, with_voxels=False)
        reps += 1
    dt = (time.perf_counter() - t0) / reps
    est = 6.0 * dt      # 6 cameras; the sparse/dense conv stacks are NOT included (they dominate on CPU)
    return {"value": 1.0 / est, "unit": "samples/s", "cores": 1, "kind": "port",
            "sample": "oracle C, 1 thread, BEV-extraction ops only: 1 of 6 cameras of bev_pool "
                      "fwd+bwd (78848 pts x 256 ch) + 1-sweep (30k pt) voxelize+mean, x6 -> upper "
                      "bound on the CPU samples/s of a full step (conv trunks excluded)"}


def main():
    args = parse()
    if os.environ.get("UD_FAULT_DUMP"):           # debugging aid: python stacks of a stuck rank after N seconds
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["UD_FAULT_DUMP"]), exit=False)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback for the product path)")
    local = local % torch.cuda.device_count()
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    if world > 1:
        backend = os.environ.get("UD_DIST_BACKEND", "nccl")       # "nccl" == RCCL on ROCm
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:                                                     # functional test of the N>1 path
            dist.init_process_group(backend)
    from unidistill_amd import _lib, train
    _lib.load()
    torch.manual_seed(1234)          # identical initial weights on every rank
    wl = WORKLOADS[args.workload]
    if wl["kind"] == "distill":
        step = train.DistillStep(args.workload)
        batch = train.synthetic_batch(device, args.batch, rank=rank, sweeps=wl["sweeps"])
    else:
        step = train.DetectStep(args.workload)
        batch = train.synthetic_batch(device, args.batch, rank=rank, sweeps=wl["sweeps"],
                                      with_imgs=args.workload != "lidar",
                                      with_points=args.workload != "camera")
    ac = torch.bfloat16 if args.autocast == "bf16" else None
    trainer = train.Trainer(step, device=device, autocast_dtype=ac, channels_last=not args.nchw)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        trainer.step(batch)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = trainer.step(batch)
    barrier()
    dt = time.perf_counter() - t0
    loss = float(out["loss"].item())
    if world > 1:
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    # the MFMA leg takes extra training steps: under DDP those are collective, so EVERY rank runs it
    mfma = None
    if not args.no_roofline and ac is not None:
        mfma = mfma_leg(trainer, batch)
    if rank == 0:
        samples = args.batch * world * args.steps
        line = {
            "metric": "distill-train samples/sec (nuScenes frame); bev_pool+voxelize HBM GB/s",
            "value": samples / dt, "unit": "samples/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None,
            "dtype": "f32" if ac is None else "bf16", "data": "synthetic",
            "config": {"workload": f"{args.workload}: student+teacher distillation step, 6 cams 256x704, "
                                   f"{30000 * wl['sweeps']}-pt cloud, 40 GT boxes, fwd+bwd+AdamW"
                       if wl["kind"] == "distill" else f"{args.workload} detector training step",
                       "batch_per_gpu": args.batch, "global_batch": args.batch * world,
                       "parallelism": f"dp{world}", "final_loss": loss,
                       "precision": "fp32 everywhere" if ac is None else
                       "bf16 operands / fp32 accumulate on dense convs, BatchNorm chains, head tail and the teacher's sparse convs (HIP MFMA kernels incl. 1x1 convs and all 3x3 / 1x1 weight gradients; libraries for strided / transposed convs and small-map 1x1 GEMMs); fp32 voxelize/splat/losses; fp32 master weights",
                       "layout": "NCHW" if args.nchw else "channels-last dense convs",
                       "executor": "eager+DDP"},
        }
        if not args.no_roofline:
            line["roofline"] = roofline_leg(device, 1)
            if mfma is not None:
                line["roofline_mfma"] = mfma
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline_leg()
        print(json.dumps(line))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
