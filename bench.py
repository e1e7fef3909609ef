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
    roofline_mfma_f32: the fp32 3x3 conv kernel family of the HEADLINE step (peak 157.3 TFLOP/s);
    roofline_spconv: the fp32 sparse encoder pass (pairs / flops / bytes per layer);
    roofline_mfma: the 3x3 conv kernel family inside the bf16 step (labelled second);
    HBM-traffic figures come from profiles/traffic.json (PMC passes, tools/make_profiles.sh), never from constants here;
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
if "--nchw" not in sys.argv:
    os.environ.setdefault("UD_STRICT", "1")    # no silent fall-through to a library convolution / GEMM in what is timed
for _p in (ROOT, PKG):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def profile_figures():
    """profiles/traffic.json: HBM bytes per launch from the PMC counter passes (FETCH_SIZE x 2 on gfx950 + WRITE_SIZE, separate
    passes as MI355X_MICROARCH.md prescribes) and the MFMA-only loop rates, written by tools/make_profiles.sh.  Counters
    cannot be read from inside this process; a missing file or a shape mismatch yields None (never a stale constant)."""
    try:
        return json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
    except Exception:
        return {}

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
    op_ms = op_dirty_ms = 0.0
    scrub.zero_()
    scrub64 = scrub.view(torch.int64)
    # three passes: the op as a whole WITHOUT the per-kernel event brackets (they add ~4 us to each of its launches), evicting
    # feat from the Infinity Cache first with a READ of 512 MB (clean lines: honest HBM reads, nothing else in flight);
    # the same after a 512 MB memset instead (rounds 1-3's protocol: the memset's dirty lines are written back WHILE the op
    # runs -- reported as op_avg_us_dirty_scrub); then the launches with the brackets for the kernel's own duration
    k_each = []
    _lib.prof_read("bev_pool.k_pool")
    for mode in ("clean", "dirty", "prof"):
        _lib.prof_enable(mode == "prof")
        for _ in range(reps):
            if mode == "dirty":
                scrub.zero_()
            else:
                scrub64.sum()
            e0.record()
            bp._pool_fwd(geom, feat, out, pos, B, N, C, nx, ny, 1, bp.POOL_OVERWRITE)
            e1.record()
            torch.cuda.synchronize()
            if mode == "clean":
                op_ms += e0.elapsed_time(e1)
            elif mode == "dirty":
                op_dirty_ms += e0.elapsed_time(e1)
            else:
                k_each.append(_lib.prof_read("bev_pool.k_pool")[0] * 1e3)      # this launch alone (the spread over a box's clocks)
    _lib.prof_enable(False)
    k_ms, k_calls = sum(k_each) * 1e-3, len(k_each)
    alg = B * N * (12 + C * 4 + 12) + B * ny * nx * C * 4          # SURVEY 8d bev_pool fwd row
    k_us = k_ms / max(k_calls, 1) * 1e3
    # SURVEY 8d's other row, reported separately: the fused lift + splat the TRAINING STEP runs (ud_lss_depth_ctx +
    # ud_lss_splat_fwd: softmax / context split, list building, k_pool<SrcLift>; the [B, N, C] tensor never exists),
    # 45.07 MB algorithmic per 6-camera sample, L2-bound
    from unidistill_amd.ops import lss as lss_ops
    D = N // (6 * 16 * 44)
    dfeat = torch.randn(B * 6, D + C, 16, 44, device=device)
    bins = geom.contiguous()
    fused_ms = 0.0
    with torch.no_grad():
        for i in range(reps + 2):
            scrub64.sum()
            e0.record()
            lss_ops.lift_splat(dfeat, bins, B, 6, D, C, nx, ny, 1)
            e1.record()
            torch.cuda.synchronize()
            if i >= 2:
                fused_ms += e0.elapsed_time(e1)
    fused_us = fused_ms / reps * 1e3
    fused_alg = B * 6 * (D + C) * 16 * 44 * 4 + B * N * 12 + B * ny * nx * C * 4
    pf = profile_figures().get("bev_pool.k_pool", {})
    traffic = pf.get("hbm_bytes") if pf.get("shape") == {"C": C, "nx": nx, "ny": ny, "N": N, "B": B} else None
    achieved = alg / (k_us * 1e-6) / 1e9
    in_grid = float((pos[..., 0] >= 0).float().mean())
    return {"bound": "hbm", "kernel": "bev_pool.k_pool (ud_bev_pool_fwd, reference op boundary)",
            "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
            # `frac` prices ALL B*N feature rows (SURVEY 8d's algorithmic bytes); the kernel only reads the in-grid ones,
            # so its real HBM throughput is frac_counter_bytes (PMC bytes / time), NOT frac: do not read frac as bandwidth
            "in_grid_fraction": in_grid,
            "frac_meaning": "algorithmic bytes (all %d rows, %.0f %% of them in-grid and actually read) / kernel time / 8 TB/s; "
                            "HBM throughput of the kernel itself = frac_counter_bytes" % (B * N, 100 * in_grid),
            "avg_kernel_us": k_us, "kernel_us_min": min(k_each), "kernel_us_max": max(k_each), "launches": k_calls,
            "algorithmic_bytes_per_launch": alg,
            "lift_splat_fused_us": fused_us, "lift_splat_fused_GBps": fused_alg / (fused_us * 1e-6) / 1e9,
            "lift_splat_fused_algorithmic_bytes": fused_alg,
            "op_avg_us": op_ms / reps * 1e3, "op_GBps": alg / (op_ms / reps * 1e-3) / 1e9,
            "op_avg_us_dirty_scrub": op_dirty_ms / reps * 1e3,
            # whole op (memset of the cell counts + k_bin + k_pool) against the same algorithmic bytes
            "frac_op": alg / (op_ms / reps * 1e-3) / 1e9 / HBM_PEAK_GBS,
            # the kernel against the HBM bytes it really moves (PMC traffic below: out-of-grid rows are never read)
            "frac_counter_bytes": (traffic / (k_us * 1e-6) / 1e9 / HBM_PEAK_GBS) if traffic else None,
            # HBM bytes per launch from the PMC passes of the same op at the same shape (profiles/traffic.json)
            "traffic": traffic, "traffic_source": pf.get("source") if traffic else None,
            "note": "measured after the timed region; HIP events bracket each launch, so avg_kernel_us carries "
                    "the ~6 us dispatch latency that rocprofv3's kernel duration (profiles/) does not; op_avg_us is "
                    "timed in a separate pass without those brackets, feat evicted from the Infinity Cache by a 512 MB read "
                    "(op_avg_us_dirty_scrub: by a 512 MB memset, whose write-back then overlaps the op); the "
                    "training step itself uses the fused lift+splat (no [B,N,C] tensor), see DESIGN.md"}


MFMA_PEAK_TFLOPS = 2500.0        # MI355X_MICROARCH.md: dense bf16 MFMA peak (no sparsity)
MFMA_PEAK_TFLOPS_F32 = 157.3     # MI355X_MICROARCH.md: fp32 matrix peak (v_mfma_f32_16x16x4_f32: 64 FLOP/clk/SIMD)


def mfma_leg(trainer, batch, fp32, steps=3):
    """The dense trunk / head / image-branch 3x3 convolutions of a training step (forward + data gradient: the largest
    kernel family of the step) -- ud_conv3x3_nhwc_f32 for the fp32 HEADLINE step, ud_conv3x3_nhwc_bf16 for the bf16 leg.
    Two measurements of the SAME launches:
      * replay (-> achieved / frac): every 3x3 launch of one training step is logged (shape, direction) and the whole
        list is replayed back to back inside ONE HIP-event bracket -- per-launch kernel time without the ~6 us that an
        event pair around a single short launch adds; this is the figure rocprofv3's kernel durations agree with;
      * in_step_events: HIP events around every launch inside real training steps (other streams running, cold L2,
        event overhead included) -- the pessimistic bound."""
    from unidistill_amd import _lib
    from unidistill_amd.ops import conv2d as c16, conv2d_f32 as c32
    c2 = c32 if fp32 else c16
    prof_names = ("conv2d.k_conv3x3_f32", "conv2d.k_conv3x3_wino_f32", "conv2d.k_conv3x3_wino4_f32") if fp32 else ("conv2d.k_conv3x3",)
    peak = MFMA_PEAK_TFLOPS_F32 if fp32 else MFMA_PEAK_TFLOPS
    c2.FLOP_COUNTER = [0]
    for nm in prof_names + ("conv2d.k_wgrad_f32", "conv2d.k_wgrad_wino_f32", "conv2d.k_wgrad_1x1_f32"):
        _lib.prof_read(nm, reset=True)
    _lib.prof_enable(True)
    for _ in range(steps):
        trainer.step(batch)
    torch.cuda.synchronize()
    _lib.prof_enable(False)
    ms, calls = (sum(v) for v in zip(*[_lib.prof_read(nm) for nm in prof_names]))
    flops, c2.FLOP_COUNTER = c2.FLOP_COUNTER[0], None
    if not calls or ms <= 0:
        return None
    c2.SHAPE_LOG = []
    trainer.step(batch)
    torch.cuda.synchronize()
    log, c2.SHAPE_LOG = c2.SHAPE_LOG, None
    dev = torch.device("cuda", torch.cuda.current_device())
    g = torch.Generator(device=dev).manual_seed(3)
    dt = torch.float32 if fp32 else torch.bfloat16
    ops, cache = [], {}
    for (B, cin, H, W, cout, rev) in log:
        key = (B, cin, H, W, cout)
        if key not in cache:
            x = torch.randn(B, cin, H, W, device=dev, generator=g).to(dt).contiguous(memory_format=torch.channels_last)
            w = (torch.randn(cout, 3, 3, cin, device=dev, generator=g) * 0.02).to(dt)
            # fp32: parameter-shaped filters [Cout, Cin, 3, 3] of the layer (the launcher picks the transform / tap order)
            wp = {r: (torch.randn((cin, cout, 3, 3) if r else (cout, cin, 3, 3), device=dev, generator=g) * 0.02) for r in (False, True)} if fp32 else None
            cache[key] = (x, w, wp)
        ops.append((cache[key], cout, rev))

    def replay():
        for (x, w, wp), cout, rev in ops:
            if fp32:
                c32._launch3(x, wp[rev], transposed=rev)
            else:
                c16._launch(x, w, cout, reverse_taps=rev)
    replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 3 if fp32 else 5
    e0.record()
    for _ in range(reps):
        replay()
    e1.record()
    torch.cuda.synchronize()
    rp_ms = e0.elapsed_time(e1) / reps
    rp_flops = sum(2 * B * H * W * cout * 9 * cin for (B, cin, H, W, cout, _) in log)
    achieved = rp_flops / (rp_ms * 1e-3) / 1e12
    # fp32: launches whose map fills the tile blocks run as Winograd F(2x2,3x3) -- 16 instead of 36 multiplications per 2 x 2
    # outputs: `achieved` stays ALGORITHMIC (direct-form) flops / time as the contract defines it, `executed` is what the MFMA
    # pipe actually ran
    def mult_frac(H, W, cin, cout):      # multiplications the routed kernel executes / the direct form's
        if not fp32:
            return 1.0
        if c32.wino4_pays(H, W, cin, cout):
            return 36.0 / 144.0          # F(4x4,3x3): 36 per 4 x 4 outputs
        return (16.0 / 36.0) if c32.wino_pays(H, W, cin, cout) else 1.0
    ex_flops = sum(2 * B * H * W * cout * 9 * cin * mult_frac(H, W, cin, cout) for (B, cin, H, W, cout, _) in log)
    n_wino = sum(1 for (B, cin, H, W, cout, _) in log if mult_frac(H, W, cin, cout) < 1.0)
    n_wino4 = sum(1 for (B, cin, H, W, cout, _) in log if fp32 and c32.wino4_pays(H, W, cin, cout))
    executed = ex_flops / (rp_ms * 1e-3) / 1e12
    in_step = flops / (ms * 1e-3) / 1e12
    if fp32:                                   # in-step events: priced by executed flops as well
        in_step *= ex_flops / rp_flops
    pf = profile_figures()
    loop = pf.get("mfma_only_loop_tflops", {})
    instr = "v_mfma_f32_16x16x4_f32" if fp32 else "v_mfma_f32_16x16x32_bf16"
    out = {"bound": "mfma",
           "kernel": ("conv2d_f32_wino4.k_conv3x3_wino4_f32 / conv2d_f32_wino.k_conv3x3_wino_f32 / conv2d_f32.k_conv_f32_taps "
                      "(ud_conv3x3_wino4_nhwc_f32, ud_conv3x3_wino_nhwc_f32, ud_conv3x3_nhwc_f32"
                      if fp32 else "conv2d.k_conv3x3_taps (ud_conv3x3_nhwc_bf16") + ": BEV trunk, head, ResNet 3x3 convs; fwd + dgrad)",
           # fp32: achieved / frac = the flops the MFMA pipe EXECUTES (Winograd launches run 16/36 -- F(2x2) -- or 36/144 -- F(4x4) --
           # of the direct-form multiplications) / time: a roofline fraction, always <= 1; the direct-form rate is kept beside it
           "achieved": executed, "peak": peak, "unit": "TFLOP/s", "frac": executed / peak, "dtype": "f32" if fp32 else "bf16",
           "launches": len(log), "avg_kernel_us": rp_ms / len(log) * 1e3, "algorithmic_flops_per_step": rp_flops,
           "kernel_ms_per_step": rp_ms, "traffic": None,
           "executed": ({"winograd_launches": n_wino, "of_them_F4x4": n_wino4, "flops_per_step": ex_flops, "TFLOP/s": executed,
                         "frac_of_peak": executed / peak} if fp32 else None),
           "algorithmic_equivalent": ({"TFLOP/s": achieved, "x_matrix_peak": achieved / peak,
                                       "note": "direct-form flops (2 B H W Cout 9 Cin) of the same launches / time: what a direct "
                                               "convolution would have to sustain to match; NOT a roofline fraction (Winograd "
                                               "F(2x2,3x3) executes 16/36, F(4x4,3x3) 36/144 of these multiplications)"} if fp32 else None),
           # what an MFMA-only loop sustains on the same machine with the instruction this kernel uses (tools/mfma_peak.hip;
           # profiles/traffic.json): informational, `frac` stays priced against the guide's nominal peak
           "mfma_only_loop": ({"instruction": instr, "TFLOP/s": loop[instr], "frac_of_loop": executed / loop[instr],
                               "source": loop.get("source")} if instr in loop else None),
           "in_step_events": {"achieved": in_step, "frac": in_step / peak, "launches": calls,
                              "avg_kernel_us": ms / calls * 1e3, "kernel_ms_per_step": ms / steps,
                              "algorithmic_flops_per_step": flops / steps},
           "note": "achieved = flops of one training step's %d conv3x3 launches / their time replayed back to back in one "
                   "HIP-event bracket (%d repetitions); in_step_events = HIP events around every launch inside %d real "
                   "steps (adds ~6 us per launch and the second stream's contention)" % (len(log), reps, steps)}
    if not fp32:
        t = pf.get("conv3x3_bf16_trunk_256_128_180x180x4")
        out["traffic_trunk_256_128_180x180x4"] = t      # PMC bytes of ONE shape (the replay mixes 65)
        out["library_gemm_reference"] = library_gemm_reference(trainer.device)
    else:
        # the fp32 weight gradients of the same step (ud_conv3x3_wgrad_nhwc_f32 + ud_conv1x1_wgrad_mapped_nhwc_f32): in-step HIP events
        w3, n3 = _lib.prof_read("conv2d.k_wgrad_f32")
        ww, nw = _lib.prof_read("conv2d.k_wgrad_wino_f32")
        w1, n1 = _lib.prof_read("conv2d.k_wgrad_1x1_f32")
        out["weight_gradients_in_step_events"] = {"k_wgrad_f32": {"ms_per_step": w3 / steps, "launches_per_step": n3 / steps},
                                                  "k_wgrad_wino_f32": {"ms_per_step": ww / steps, "launches_per_step": nw / steps},
                                                  "k_wgrad_1x1_f32": {"ms_per_step": w1 / steps, "launches_per_step": n1 / steps}}
    return out


def library_gemm_reference(device):
    """What the vendor's own large bf16 GEMM sustains on THIS box with random operands (8192^3 through `torch @`, hipBLASLt): the bf16
    MFMA kernels run into the 1 400 W package limit and the shader clock drops to ~2.0 GHz (profiles/r04_conv_bf16.md, section 7),
    so this -- not 2.5 PF -- is what a GEMM-shaped kernel reaches here.  A reference point only: nothing on the product path calls it,
    `frac` stays priced against the guide's nominal peak."""
    a = torch.randn(8192, 8192, device=device, dtype=torch.bfloat16)
    for _ in range(20):
        a @ a
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 200
    e0.record()
    for _ in range(n):
        a @ a
    e1.record()
    torch.cuda.synchronize()
    tf = 2 * 8192 ** 3 * n / (e0.elapsed_time(e1) * 1e-3) / 1e12
    return {"TFLOP/s": tf, "frac_of_peak": tf / MFMA_PEAK_TFLOPS, "what": "torch @ (hipBLASLt), 8192^3 bf16, random operands, 200 calls"}


def spconv_leg(device, batch):
    """SURVEY 8d sparse-conv row: one fp32 pass of the LiDAR encoder (VoxelResBackBone8x, spconv_backbone.py:259-341) on the
    batch's clouds -- per layer rows, pairs (measured from the rulebook), flops = 2 pairs Cin Cout, bytes = pairs (Cin + Cout)
    4 + |W| -- with the pass's `spconv.k_conv` launches timed by HIP events.  MFMA-bound fraction reported on the 128-channel
    layers (v_mfma_f32_16x16x4_f32, peak 157.3), HBM gather/scatter fraction on the narrower ones."""
    from unidistill_amd import _lib, config as C
    from unidistill_amd.layers.lidar import LidarEncoder
    from unidistill_amd.ops import spconv as sp
    torch.manual_seed(7)
    enc = LidarEncoder(C.LIDAR_ENCODER).to(device).eval()
    pts = [p for p in batch["points"]]
    with torch.no_grad():
        for _ in range(2):
            enc(pts)
        sp.CONV_LOG = []
        enc(pts)
        log, sp.CONV_LOG = sp.CONV_LOG, None
        pairs = {}
        for nbr, _, _, _ in log:
            if id(nbr) not in pairs:
                pairs[id(nbr)] = int((nbr >= 0).sum())
        _lib.prof_read("spconv.k_conv", reset=True)      # HIP events around every conv launch of one whole pass
        _lib.prof_enable(True)
        enc(pts)
        torch.cuda.synchronize()
        _lib.prof_enable(False)
        tot_ms, tot_n = _lib.prof_read("spconv.k_conv")
        # per-layer split: replay each layer's conv alone (same rulebook, random features), HIP events around 5 launches
        layers = []
        g = torch.Generator(device=device).manual_seed(1)
        for i, (nbr, cin, cout, kind) in enumerate(log):
            M, K = nbr.shape
            feat = torch.randn(int(nbr.max().item()) + 1, cin, device=device, generator=g)
            w = torch.randn(cout, K, cin, device=device, generator=g) * 0.05
            for _ in range(2):
                sp._conv(feat, nbr, w, (K * cin, cin, 1), False, None, cin, cout, 0)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                sp._conv(feat, nbr, w, (K * cin, cin, 1), False, None, cin, cout, 0)
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / 5 * 1e3
            pr = pairs[id(nbr)]
            fl = 2.0 * pr * cin * cout
            by = pr * (cin + cout) * 4.0 + cout * K * cin * 4.0
            layers.append({"rows": M, "K": K, "cin": cin, "cout": cout, "pairs": pr, "flops": fl, "bytes": by, "us": us,
                           "TFLOP/s": fl / us / 1e6, "GB/s": by / us / 1e3})
    wide = [l for l in layers if l["cin"] == 128 and l["cout"] == 128]
    narrow = [l for l in layers if l["cin"] <= 64]
    fl_w, us_w = sum(l["flops"] for l in wide), sum(l["us"] for l in wide)
    by_n, us_n = sum(l["bytes"] for l in narrow), sum(l["us"] for l in narrow)
    fl_all, us_all = sum(l["flops"] for l in layers), sum(l["us"] for l in layers)
    ach = fl_w / us_w / 1e6 if us_w else 0.0
    return {"bound": "mfma", "kernel": "spconv_conv.k_conv_dma_f32<128,128> (ud_spconv_conv, fp32: the 128-channel layers of "
                                       "VoxelResBackBone8x; narrower layers are gather/scatter-bound, see hbm_layers)",
            "achieved": ach, "peak": MFMA_PEAK_TFLOPS_F32, "unit": "TFLOP/s", "frac": ach / MFMA_PEAK_TFLOPS_F32,
            "launches": len(wide), "avg_kernel_us": us_w / max(len(wide), 1), "traffic": None,
            "hbm_layers": {"bound": "hbm", "achieved": by_n / us_n / 1e3 if us_n else 0.0, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                           "frac": (by_n / us_n / 1e3 / HBM_PEAK_GBS) if us_n else 0.0, "launches": len(narrow),
                           "note": "layers with Cin <= 64: SURVEY 8d bytes = pairs (Cin + Cout) 4 + |W| (gather + scatter traffic "
                                   "of a pair-list formulation; the output-stationary kernel writes each row once)"},
            "encoder_pass": {"conv_launches": len(layers), "pairs": sum(l["pairs"] for l in layers), "flops": fl_all,
                             "sum_of_layer_us": us_all, "TFLOP/s": fl_all / us_all / 1e6,
                             "in_pass_events": {"ms": tot_ms, "launches": tot_n}},
            "layers": layers,
            "note": "clouds of the bench batch (%d x %d points); per-layer times: each layer's launch replayed alone on its real "
                    "rulebook (HIP events around 5 launches); in_pass_events: HIP events around every launch inside one encoder pass"
                    % (len(pts), pts[0].shape[0])}


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
        m = torch.empty(B + 2, dtype=torch.int32, device=device)
        # what the product wrapper picks (ops/voxelize.py): below 160 k points the three-launch atomic hash on its own clean
        # workspace (algo 3 after one algo-2 call), else the hash partition + LDS sort (algo 0: no global atomics; the
        # synthetic clouds never overflow a partition)
        small = B * N < 160000
        ALGO = 3 if small else 0
        vs, rg, st = _f3(syn.VOXEL_SIZE), _f3(syn.POINT_CLOUD_RANGE), _lib.stream_of(pts)

        def run(algo=ALGO):
            _lib.check(lib.ud_voxelize(_lib.ptr(pts), B, N, F, vs, rg, P, maxM, _lib.ptr(vox), _lib.ptr(coords),
                                       _lib.ptr(num), None, _lib.ptr(m), _lib.ptr(ws), ws.numel(), algo, st), "ud_voxelize")
        if small:
            run(2)
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
        for k in (("k_insert2", "k_first_assign", "k_gather") if small else ("k_partition", "k_bucket", "k_flags", "k_rows", "k_gather")):
            ms, n = _lib.prof_read("voxelize." + k)
            kern[k] = ms / max(n, 1) * 1e3
        M = int(m[B])
        if int(m[B + 1]) != 0:                 # a partition overflowed: algo 0 left the voxelization incomplete
            raise RuntimeError("voxelize_leg: algo 0 overflowed a partition on the synthetic cloud; the timed op is not a "
                               "complete voxelization (the product wrapper would repeat with algo 1)")
        alg = B * N * F * 4 + M * (P * F * 4 + 12 + 4)          # N*20 + M*216 bytes
        dom = max(kern, key=kern.get)
        cases.append({"points": B * N, "voxels": M, "algo": ALGO, "algorithmic_bytes": alg, "op_us": op_us,
                      "op_GBps": alg / op_us / 1e3, "frac_op": alg / op_us / 1e3 / HBM_PEAK_GBS,
                      "kernel_us": kern, "dominant_kernel": "voxelize." + dom})
    big = cases[-1]
    pv = profile_figures().get("voxelize", {})
    sh = pv.get("shape", {})
    vox_traffic = pv.get("hbm_bytes") if abs(big["points"] - sh.get("points", -10 ** 9)) <= sh.get("tolerance", 0) else None
    return {"bound": "hbm", "kernel": "ud_voxelize (reference op boundary: voxels[M,10,5] + coords + num), whole op",
            "achieved": big["op_GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": big["frac_op"],
            # PMC bytes of the four kernels at this size (profiles/r02_pmc_voxelize.md, max = the 1.19 M-point case: FETCH_SIZE x 2
            # (gfx950 correction) + WRITE_SIZE, KB): 4.7x the algorithmic bytes -- the random 8-byte hash-table accesses of
            # k_insert / k_first move whole sectors
            "traffic": vox_traffic, "traffic_source": pv.get("source") if vox_traffic else None, "cases": cases,
            "traffic_ratio": (vox_traffic / big["algorithmic_bytes"]) if vox_traffic else None,
            "small_cloud_us": cases[0]["op_us"],
            "note": "op-level; cases[1] (headline of this object): algo 0 (hash partition + per-partition LDS hash / counting "
                    "sort: no global atomics; 6 launches); cases[0]: the 30 k cloud on the three-launch atomic hash with a "
                    "self-cleaning workspace (algo 3).  Per-kernel times are HIP-event brackets incl. ~6 us dispatch.  "
                    "Random-access bound (gathers of 20-byte points, 4-16-byte scattered stores), not streaming-bound: see DESIGN.md"}


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
    import oracle
    prev = torch.get_num_threads()
    torch.set_num_threads(ncores)
    os.environ["OMP_NUM_THREADS"] = str(ncores)
    was = oracle.use_openmp(True)   # lift + voxel pooling fwd / bwd from the OpenMP build of the oracle source
    t_all = timed(10, 2)            # 2 warm-up + 10 timed iterations
    oracle.use_openmp(False)
    t_scalar = timed(3, 1)          # the same with the scalar checker inside (rounds 1-4's figure)
    torch.set_num_threads(1)
    t_one = timed(3, 1)             # one thread: 1 + 3 (a step takes ~7 s there; keeps the default run within minutes)
    torch.set_num_threads(prev)
    oracle.use_openmp(was)
    # the product path on the same configuration (camera detector, 1 camera, batch 1, fp32, fwd+bwd+AdamW)
    torch.manual_seed(1234)
    tr = train.Trainer(train.DetectStep("camera"), device=device, channels_last=True)
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
            "label": "port: torch CPU ops on all cores + the OpenMP build of the oracle source for the lift and voxel pooling "
                     "forward / backward (oracle/Makefile: libud_oracle_omp.so, bit-identical to the scalar checker); the frustum "
                     "geometry stays numpy",
            "scalar_oracle_inside": {"value": 1.0 / sorted(t_scalar)[len(t_scalar) // 2],
                                     "seconds_per_step": sorted(t_scalar)[len(t_scalar) // 2], "iterations": 3, "warmup": 1,
                                     "note": "same step with the single-thread checker build inside (rounds 1-4)"},
            "sample": "BASELINE configs[0]: camera-only student, 1 camera 256x704, batch 1, random weights, "
                      "forward+backward; torch CPU ops (image branch, BEV trunk, head, target assignment, loss) + "
                      "CPU oracle (geometry, lift, voxel pooling fwd/bwd); median of 10 iterations after 2 warm-ups on the "
                      f"{ncores} cores this process may use (affinity mask capped by the cgroup CPU quota; the host has "
                      f"{os.cpu_count()}); one thread: median of 3 after 1 warm-up",
            "seconds_per_step_all_cores": {"median": med, "min": min(t_all), "iterations": 10, "warmup": 2},
            "one_thread": {"value": 1.0 / sorted(t_one)[1], "seconds_per_step": sorted(t_one)[1], "min": min(t_one),
                           "iterations": 3, "warmup": 1},
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


def host_enqueue_leg(workload, batch, bf16):
    """Host-side cost of one training step: CPU time of the threads that ENQUEUE it -- the main thread (forward, losses,
    optimizer) and the autograd engine's thread (backward); they run one after the other, so their sum is the host's critical
    path per step: a step is host-bound when it approaches ms_per_step (8 ranks on one host: this, not xGMI, bounds the
    scaling).  Measured in a CHILD process (tools/host_threads.py --json) because the runtime's synchronisation has to be
    switched from spinning to blocking BEFORE the first launch for the GPU waits not to count; the HIP runtime's own polling
    thread is listed separately."""
    import subprocess
    env = dict(os.environ, WL=workload, B=str(batch), AC="bf16" if bf16 else "f32")
    try:
        res = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "host_threads.py"), "--json"], env=env,
                             capture_output=True, text=True, timeout=600)
        line = [l for l in res.stdout.splitlines() if l.startswith("{")][-1]
        out = json.loads(line)
        out["note"] = ("per-thread CPU time from /proc/self/task (10 ms ticks over %d steps) in a child process; "
                       "host_enqueue_ms = main + autograd thread" % out["steps"])
        return out
    except Exception as e:          # never fail the benchmark line over the side measurement
        return {"host_enqueue_ms": None, "error": repr(e)[:200]}


def relaunch_ranks(n):
    """Run this command line as n ranks on this node (127.0.0.1 rendezvous on a free port); rank 0 prints the JSON line."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    raise SystemExit(subprocess.call(cmd, env=env))


def frac_violations(obj, path="line"):
    """Every `frac` in the line must be a roofline fraction in [0, 1]; returns the offenders (the line is printed first)."""
    bad = []
    try:
        assert_fracs(obj, path)
    except AssertionError as e:
        bad.append(str(e))
    return bad


def assert_fracs(obj, path="line"):
    """Every emitted `frac` is a roofline fraction: <= 1 by construction (a larger value means the timed kernel does not do
    the counted work)."""
    if isinstance(obj, dict):
        for k, v in obj.items():
            if k == "frac" and v is not None:
                assert 0.0 <= v <= 1.0, f"{path}.frac = {v}: not a roofline fraction"
            assert_fracs(v, f"{path}.{k}")
    elif isinstance(obj, (list, tuple)):
        for i, v in enumerate(obj):
            assert_fracs(v, f"{path}[{i}]")


PRECISION_NOTE = {
    None: "fp32 everywhere (the reference's arithmetic): hand-written HIP for voxelize / sparse convs (fp32 MFMA) / "
          "lift-splat / target assignment / losses / BatchNorm+ReLU chains / head tail (grouped fp32 kernels), every "
          "convolution of the step but the frozen 7x7 stem -- 3x3, 1x1, strided, transposed: forward, data and weight "
          "gradient -- on fp32 MFMA kernels (v_mfma_f32_16x16x4_f32), deterministic",
    torch.bfloat16: "bf16 operands / fp32 accumulate on dense convs, BatchNorm chains, head tail and the sparse convs "
                    "(HIP MFMA kernels for every convolution of the step -- 3x3, 1x1, strided, transposed: forward, data "
                    "and weight gradients -- except the frozen 7x7 stem); fp32 voxelize/splat/losses; fp32 master weights",
}


def main():
    args = parse()
    if os.environ.get("UD_FAULT_DUMP"):           # debugging aid: python stacks of a stuck rank after N seconds
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["UD_FAULT_DUMP"]), exit=False)
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # `python bench.py --gpus N` on its own: one process per GPU, as the reference's `--gpus N` (Lightning
        # accelerator="ddp", exps/base_cli.py:40-58, README.md:92) -- re-launch this command line under torch.distributed.run
        return relaunch_ranks(args.gpus)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: the launch does not match the request")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback for the product path)")
    backend = os.environ.get("UD_DIST_BACKEND", "nccl") if world > 1 else "none"    # "nccl" == RCCL on ROCm
    if backend == "nccl" and world > torch.cuda.device_count():
        raise SystemExit(f"bench.py: {world} RCCL ranks need {world} GPUs, {torch.cuda.device_count()} visible "
                         "(UD_DIST_BACKEND=gloo shares devices: functional test only)")
    local = local % torch.cuda.device_count()
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    if world > 1:
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
    from unidistill_amd.ops import wgrad_stream
    wgrad_state = wgrad_stream.state()
    if wgrad_state == "ddp":
        wgrad_state = "ddp (%d gradients written into bucket views, %d inline while the views settled)" % (
            wgrad_stream.STATS["ddp_direct"], wgrad_stream.STATS["ddp_inline"])
    # second precision + the MFMA legs take extra (collective) training steps: EVERY rank runs them
    bf16, mfma, mfma32 = None, None, None
    if ac is None and not args.no_roofline:
        mfma32 = mfma_leg(trainer, batch, fp32=True)
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
            mfma = mfma_leg(trainer16, batch, fp32=False)
        del trainer16
    elif ac is not None and not args.no_roofline:
        mfma = mfma_leg(trainer, batch, fp32=False)
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
                       "executor": "eager+DDP", "strict_no_library_fallthrough": os.environ.get("UD_STRICT") == "1",
                       # weight gradients on their own HIP stream (ops/wgrad_stream.py): "on", "off", or "ddp" (kept under
                       # DistributedDataParallel by writing into the bucket views + a per-bucket join in a communication hook)
                       "wgrad_stream": wgrad_state,
                       "ranks": dist.get_world_size() if world > 1 else 1,
                       "dist_backend": dist.get_backend() if world > 1 else "none (single process)"},
        }
        if bf16 is not None:
            line["bf16_mixed_precision"] = bf16
        if not args.no_roofline:
            line["roofline"] = roofline_leg(device, 1)
            line["roofline_voxelize"] = voxelize_leg(device)
            # flat copies inside `roofline` (BASELINE.json's metric names "bev_pool+voxelize HBM GB/s"; the driver's parsed record
            # keeps the `roofline` object)
            rv = line["roofline_voxelize"]
            line["roofline"].update(voxelize_frac=rv["frac"], voxelize_GBps=rv["achieved"], voxelize_traffic_ratio=rv["traffic_ratio"],
                                    voxelize_small_cloud_us=rv["small_cloud_us"])
            if mfma32 is not None:
                line["roofline_mfma_f32"] = mfma32
                line["roofline"].update(trunk_mfma_f32_frac_executed=mfma32["frac"],
                                        trunk_direct_equiv_TFLOPs=mfma32["algorithmic_equivalent"]["TFLOP/s"])
            if "points" in batch:
                line["roofline_spconv"] = spconv_leg(device, batch)
                line["roofline"].update(spconv_frac=line["roofline_spconv"].get("frac"))
            if mfma is not None:
                line["roofline_mfma"] = mfma
                line["roofline"].update(trunk_mfma_bf16_frac=mfma["frac"])
        if world == 1 and not args.no_roofline and wl["kind"] == "distill":
            host = host_enqueue_leg(args.workload, args.batch, False)
            line["host_enqueue_ms"], line["host_enqueue"] = host.get("host_enqueue_ms"), host
            if bf16 is not None:
                line["bf16_mixed_precision"]["host_enqueue"] = host_enqueue_leg(args.workload, args.batch, True)
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline_leg(device)
        bad = frac_violations(line)
        if bad:
            line["frac_violations"] = bad
        print(json.dumps(line), flush=True)     # the measurements are emitted even if a fraction is out of range
        if bad:
            raise SystemExit("bench.py: roofline fraction(s) outside [0, 1]: " + "; ".join(bad))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
