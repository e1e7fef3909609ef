"""bench.py -- UniDistill distillation training on MI355X: one JSON line per run (driver contract).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one full distillation training step (student fwd + frozen-teacher fwd + detection and
distillation losses + backward + grad clip + AdamW) over one per-GPU batch of synthetic
nuScenes-shaped input.  The path is pure data parallel: every rank trains on its
own samples and gradients are averaged with RCCL, so `scaling` is "weak" and
value = samples of all ranks / max-over-ranks wall time.

HEADLINE = fp32, the reference's arithmetic (it has no autocast anywhere: base_cli.py:40-45); the bf16
mixed-precision step (BASELINE.json configs[4] style) is timed afterwards with its own trainer and reported
as the labelled object `bf16_mixed_precision` -- never as `value`.

After the timed region (never counted in `value`) further legs run:
  * bf16_mixed_precision (all ranks): the same workload under bf16 autocast + channels-last;
  * roofline (rank 0): the reference-boundary bev_pool forward (BASELINE.json: "bev_pool+voxelize HBM GB/s")
    at the BASELINE shape, its dominant kernel timed with HIP events on its launch stream, plus the
    op-level and counter-byte fractions; roofline_voxelize: ud_voxelize at 30 k and 4 x 300 k points;
    roofline_mfma: the 3x3 conv kernel family inside the bf16 step;
  * cpu_baseline (rank 0, N=1): BASELINE configs[0] (camera-only student, 1 camera, batch 1, fwd+bwd) end to end
    on the host cores -- torch CPU ops + the CPU oracle for the native ops (oracle/cpu_step.py), all cores and
    one thread -- with the GPU timed on the same configuration beside it.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "cvpr2023-unidistill_amd")
os.environ.setdefault("UD_RANDOM_INIT", "1")   # synthetic benchmark: random weights of the reference architecture
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
    ap.add_argument("--autocast", default="none", choices=["none", "bf16"],
                    help="precision of the HEADLINE: none = fp32 (the reference's arithmetic, default); bf16 = "
                         "mixed precision as the headline (then no second leg)")
    ap.add_argument("--no-bf16-leg", action="store_true", help="skip the bf16 mixed-precision leg")
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
    reps = 10
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    op_ms = 0.0
    # two passes: the op as a whole WITHOUT the per-kernel event brackets (they add ~4 us to each of its five launches),
    # then the same launches with them for the kernel's own duration
    for prof in (False, True):
        _lib.prof_enable(prof)
        for _ in range(reps):
            scrub.zero_()                       # evict feat from the Infinity Cache: honest HBM reads
            e0.record()
            bp._pool_fwd(geom, feat, out, pos, B, N, C, nx, ny, 1, bp.POOL_OVERWRITE)
            e1.record()
            torch.cuda.synchronize()
            if not prof:
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
            # whole op (memset + k_bin + scan + k_fill + k_pool) against the same algorithmic bytes
            "frac_op": alg / (op_ms / reps * 1e-3) / 1e9 / HBM_PEAK_GBS,
            # the kernel against the HBM bytes it really moves (PMC traffic below: out-of-grid rows are never read)
            "frac_counter_bytes": (406.8e6 / (k_us * 1e-6) / 1e9 / HBM_PEAK_GBS)
            if (C, nx, ny, N) == (256, 180, 180, 473088) else None,
            # HBM bytes per launch from the PMC passes of the same op at the same shape
            # (profiles/r01_pmc_k_pool.md: FETCH_SIZE 182401 KB x2 (gfx950) + WRITE_SIZE 32400 KB);
            # counters cannot be read from inside this process, so the committed figure is reported.
            "traffic": 406.8e6 if (C, nx, ny, N) == (256, 180, 180, 473088) else None,
            "note": "measured after the timed region; HIP events bracket each launch, so avg_kernel_us carries "
                    "the ~6 us dispatch latency that rocprofv3's kernel duration (profiles/) does not; op_avg_us is "
                    "timed in a separate pass without those brackets; the "
                    "training step itself uses the fused lift+splat (no [B,N,C] tensor), see DESIGN.md"}


MFMA_PEAK_TFLOPS = 2500.0   # MI355X_MICROARCH.md: dense bf16 MFMA peak (no sparsity)


def mfma_leg(trainer, batch, steps=3):
    """The dense trunk / head / image-branch 3x3 convolutions (ud_conv3x3_nhwc_bf16: forward + data gradient, the largest
    kernel family of the bf16 step).  Two measurements of the SAME launches:
      * replay (-> achieved / frac): every 3x3 launch of one training step is logged (shape, direction) and the whole
        list is replayed back to back inside ONE HIP-event bracket -- per-launch kernel time without the ~6 us that an
        event pair around a single short launch adds; this is the figure rocprofv3's kernel durations agree with;
      * in_step_events: HIP events around every launch inside real training steps (other streams running, cold L2,
        event overhead included) -- the pessimistic bound."""
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
    c2.SHAPE_LOG = []
    trainer.step(batch)
    torch.cuda.synchronize()
    log, c2.SHAPE_LOG = c2.SHAPE_LOG, None
    dev = torch.device("cuda", torch.cuda.current_device())
    g = torch.Generator(device=dev).manual_seed(3)
    ops, cache = [], {}
    for (B, cin, H, W, cout, rev) in log:
        key = (B, cin, H, W, cout)
        if key not in cache:
            x = torch.randn(B, cin, H, W, device=dev, generator=g).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
            w = (torch.randn(cout, 3, 3, cin, device=dev, generator=g) * 0.02).to(torch.bfloat16)
            cache[key] = (x, w)
        ops.append((cache[key], cout, rev))

    def replay():
        for (x, w), cout, rev in ops:
            c2._launch(x, w, cout, reverse_taps=rev)
    replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 5
    e0.record()
    for _ in range(reps):
        replay()
    e1.record()
    torch.cuda.synchronize()
    rp_ms = e0.elapsed_time(e1) / reps
    rp_flops = sum(2 * B * H * W * cout * 9 * cin for (B, cin, H, W, cout, _) in log)
    achieved = rp_flops / (rp_ms * 1e-3) / 1e12
    in_step = flops / (ms * 1e-3) / 1e12
    return {"bound": "mfma", "kernel": "conv2d.k_conv3x3 (ud_conv3x3_nhwc_bf16: BEV trunk, head, ResNet 3x3 convs; fwd + dgrad)",
            "achieved": achieved, "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": achieved / MFMA_PEAK_TFLOPS,
            "launches": len(log), "avg_kernel_us": rp_ms / len(log) * 1e3, "algorithmic_flops_per_step": rp_flops,
            "kernel_ms_per_step": rp_ms, "traffic": None,
            # PMC bytes of ONE shape (the replay mixes 65): FETCH_SIZE x 2 + WRITE_SIZE, profiles/r02_pmc_conv3x3.md
            "traffic_trunk_256_128_180x180x4": {"hbm_bytes": 110.6e6, "algorithmic_bytes": 100.1e6,
                                                "source": "profiles/r02_pmc_conv3x3.md"},
            # what an MFMA-only loop sustains on the same machine with the instruction this kernel uses
            # (tools/mfma_peak.hip, profiles/r02_mfma_peak.md): the nominal 2 517 is not reachable with 16x16x32
            "mfma_only_loop": {"v_mfma_f32_16x16x32_bf16": 1330.0, "v_mfma_f32_32x32x16_bf16": 2350.0, "unit": "TFLOP/s",
                               "frac_of_16x16x32_loop": achieved / 1330.0, "source": "profiles/r02_mfma_peak.md"},
            "in_step_events": {"achieved": in_step, "frac": in_step / MFMA_PEAK_TFLOPS, "launches": calls,
                               "avg_kernel_us": ms / calls * 1e3, "kernel_ms_per_step": ms / steps,
                               "algorithmic_flops_per_step": flops / steps},
            "note": "achieved = flops of one training step's %d conv3x3 launches / their time replayed back to back in one "
                    "HIP-event bracket (%d repetitions); in_step_events = HIP events around every launch inside %d real "
                    "steps (adds ~6 us per launch and the second stream's contention)" % (len(log), reps, steps)}


def voxelize_leg(device):
    """ud_voxelize at the reference op boundary (voxels[M,P,F] + coords + num):
    one 30 k-point cloud (BASELINE configs[1]) and 4 x ten-sweep clouds (configs[3]/[4], 1.2 M points)."""
    from unidistill_amd import _lib, synthetic as syn
    from unidistill_amd.ops.voxelize import _f3
    lib = _lib.load()
    cases = []
    for B, sweeps in ((1, 1), (4, 10)):
        g = syn.rng(5)
        pts = torch.from_numpy(syn.pad_clouds([syn.lidar_cloud(g, 30000, sweeps) for _ in range(B)])).to(device)
        _, N, F = pts.shape
        P, maxM = 10, 120000
        cap = lib.ud_voxelize_capacity(B, N, maxM)
        ws = _lib.workspace(device, lib.ud_voxelize_workspace_bytes(B, N, P, maxM), "voxelize")
        vox = torch.empty(cap, P, F, device=device)
        coords = torch.empty(cap, 4, dtype=torch.int32, device=device)
        num = torch.empty(cap, dtype=torch.int32, device=device)
        m = torch.empty(B + 1, dtype=torch.int32, device=device)
        vs, rg, st = _f3(syn.VOXEL_SIZE), _f3(syn.POINT_CLOUD_RANGE), _lib.stream_of(pts)

        def run():
            _lib.check(lib.ud_voxelize(_lib.ptr(pts), B, N, F, vs, rg, P, maxM, _lib.ptr(vox), _lib.ptr(coords),
                                       _lib.ptr(num), None, _lib.ptr(m), _lib.ptr(ws), ws.numel(), st), "ud_voxelize")
        for _ in range(3):
            run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 30
        e0.record()
        for _ in range(reps):
            run()
        e1.record()
        torch.cuda.synchronize()
        op_us = e0.elapsed_time(e1) / reps * 1e3
        _lib.prof_enable(True)
        for _ in range(10):
            run()
        torch.cuda.synchronize()
        _lib.prof_enable(False)
        kern = {}
        for k in ("k_insert", "k_first", "k_assign", "k_gather"):
            ms, n = _lib.prof_read("voxelize." + k)
            kern[k] = ms / max(n, 1) * 1e3
        M = int(m[B])
        alg = B * N * F * 4 + M * (P * F * 4 + 12 + 4)          # N*20 + M*216 bytes
        dom = max(kern, key=kern.get)
        cases.append({"points": B * N, "voxels": M, "algorithmic_bytes": alg, "op_us": op_us,
                      "op_GBps": alg / op_us / 1e3, "frac_op": alg / op_us / 1e3 / HBM_PEAK_GBS,
                      "kernel_us": kern, "dominant_kernel": "voxelize." + dom})
    big = cases[-1]
    return {"bound": "hbm", "kernel": "ud_voxelize (reference op boundary: voxels[M,10,5] + coords + num), whole op",
            "achieved": big["op_GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": big["frac_op"],
            # PMC bytes of the four kernels at this size (profiles/r02_pmc_voxelize.md, max = the 1.19 M-point case: FETCH_SIZE x 2
            # (gfx950 correction) + WRITE_SIZE, KB): 4.7x the algorithmic bytes -- the random 8-byte hash-table accesses of
            # k_insert / k_first move whole sectors
            "traffic": 602.7e6 if big["points"] > 1000000 else None, "cases": cases,
            "note": "op-level (memset + 5 launches); per-kernel times are HIP-event brackets incl. ~6 us dispatch. "
                    "The op is bound by random 8-byte hash-table accesses (k_insert), not by streaming bytes: see DESIGN.md"}


def usable_cores():
    """Cores this process may really use: affinity mask, capped by the cgroup CPU quota (a container that sees
    the host's core count but owns a fraction of it thrashes when torch spawns one thread per visible core)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                if q > 0:
                    n = min(n, max(1, q // int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())))
            break
        except Exception:
            continue
    return max(1, n)


def cpu_baseline_leg(device):
    """BASELINE.json configs[0] end to end on the host cores (oracle/cpu_step.py: torch CPU ops for the
    image branch / trunk / head / losses + the CPU oracle for geometry, lift and voxel pooling), all cores
    and one thread; the GPU product path on the SAME configuration is timed beside it."""
    from oracle import cpu_step
    from unidistill_amd import train
    ncores = min(usable_cores(), 64)      # torch's intra-op pool stops scaling long before that on these convs
    model, batch = cpu_step.build()

    def timed(iters, warm):
        for _ in range(warm):
            cpu_step.step(model, batch)
        ts = []
        for _ in range(iters):
            t0 = time.perf_counter()
            cpu_step.step(model, batch)
            ts.append(time.perf_counter() - t0)
        return ts
    prev = torch.get_num_threads()
    torch.set_num_threads(ncores)
    t_all = timed(3, 1)
    torch.set_num_threads(1)
    t_one = timed(1, 0)
    torch.set_num_threads(prev)
    # the product path on the same configuration (camera detector, 1 camera, batch 1, fp32, fwd+bwd+AdamW)
    torch.manual_seed(1234)
    tr = train.Trainer(train.DetectStep("camera"), device=device)
    gb = train.synthetic_batch(device, 1, ncam=1, with_points=False)
    for _ in range(3):
        tr.step(gb)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        tr.step(gb)
    torch.cuda.synchronize()
    gpu_s = (time.perf_counter() - t0) / 10
    med = sorted(t_all)[len(t_all) // 2]
    try:
        cpu_name = [l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][0]
    except Exception:
        cpu_name = "unknown"
    return {"value": 1.0 / med, "unit": "samples/s", "cores": ncores, "kind": "port",
            "sample": "BASELINE configs[0]: camera-only student, 1 camera 256x704, batch 1, random weights, "
                      "forward+backward; torch CPU ops (image branch, BEV trunk, head, target assignment, loss) + "
                      "CPU oracle (geometry, lift, voxel pooling fwd/bwd); median of 3 iterations after 1 warm-up",
            "seconds_per_step_all_cores": {"median": med, "min": min(t_all)},
            "one_thread": {"value": 1.0 / t_one[0], "seconds_per_step": t_one[0], "iterations": 1},
            "cpu_model": cpu_name,
            "gpu_same_config": {"value": 1.0 / gpu_s, "ms_per_step": gpu_s * 1e3, "dtype": "f32",
                                "note": "unidistill_amd camera detector, 1 camera, batch 1, fwd+bwd+AdamW on 1 MI355X"}}


def timed_steps(trainer, batch, args, world, device):
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
    return dt, loss


PRECISION_NOTE = {
    None: "fp32 everywhere (the reference's arithmetic): hand-written HIP for voxelize / sparse convs (fp32 MFMA) / "
          "lift-splat / target assignment / losses / BatchNorm+ReLU chains / head tail (grouped fp32 kernels), every "
          "stride-1 3x3 convolution (forward + data gradient) and the 1x1 convolutions where ours beats the library on "
          "fp32 MFMA kernels; strided / transposed / remaining 1x1 convolutions and all dense weight gradients through "
          "MIOpen fp32",
    torch.bfloat16: "bf16 operands / fp32 accumulate on dense convs, BatchNorm chains, head tail and the sparse convs "
                    "(HIP MFMA kernels for every convolution of the step -- 3x3, 1x1, strided, transposed: forward, data "
                    "and weight gradients -- except the frozen 7x7 stem); fp32 voxelize/splat/losses; fp32 master weights",
}


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
    dt, loss = timed_steps(trainer, batch, args, world, device)
    # second precision + the MFMA leg take extra (collective) training steps: EVERY rank runs them
    bf16, mfma = None, None
    if ac is None and not args.no_bf16_leg:
        del trainer
        torch.manual_seed(1234)
        step16 = train.DistillStep(args.workload) if wl["kind"] == "distill" else train.DetectStep(args.workload)
        trainer16 = train.Trainer(step16, device=device, autocast_dtype=torch.bfloat16, channels_last=not args.nchw)
        dt16, loss16 = timed_steps(trainer16, batch, args, world, device)
        bf16 = {"value": args.batch * world * args.steps / dt16, "unit": "samples/s",
                "ms_per_step": dt16 / args.steps * 1e3, "dtype": "bf16", "steps": args.steps, "warmup": args.warmup,
                "final_loss": loss16, "precision": PRECISION_NOTE[torch.bfloat16],
                "note": "same workload, batch and step as the headline under bf16 autocast + channels-last "
                        "(BASELINE.json configs[4]-style mixed precision; NOT the headline: the reference trains in fp32)"}
        if not args.no_roofline:
            mfma = mfma_leg(trainer16, batch)
    elif ac is not None and not args.no_roofline:
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
                       "precision": PRECISION_NOTE[ac],
                       "layout": "NCHW" if args.nchw else "channels-last dense convs",
                       "executor": "eager+DDP"},
        }
        if bf16 is not None:
            line["bf16_mixed_precision"] = bf16
        if not args.no_roofline:
            line["roofline"] = roofline_leg(device, 1)
            line["roofline_voxelize"] = voxelize_leg(device)
            if mfma is not None:
                line["roofline_mfma"] = mfma
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline_leg(device)
        print(json.dumps(line))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
