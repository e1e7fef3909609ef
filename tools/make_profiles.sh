#!/bin/bash
# Regenerate the rocprofv3 evidence on the GPU box (run through gpurun); summaries land in
# gpurun_out/profiles_rNN/ and are then copied to profiles/ (tracked), together with profiles/traffic.json, which
# bench.py reads for its `traffic` figures.
#   gpurun --timeout 2400 -- 'bash tools/make_profiles.sh r04'
set -u
R=${1:-r06}
OUT=$GRAFT_REPO_ROOT/gpurun_out/profiles_$R
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
T=$GRAFT_REPO_ROOT/tools
# 1. the bench command itself: kernel trace; summary restricted to the steady state (no warm-up / MIOpen find kernels)
rm -rf /tmp/p_bench; rocprofv3 --kernel-trace --stats -d /tmp/p_bench -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline > $OUT/bench_stdout.txt 2>&1
DB=$(ls -S /tmp/p_bench/*/*.db | head -1)   # the bench process itself: the LARGEST database (bench.py times the host enqueue in a child process, whose trace is a second, smaller file)
{ echo "# rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline  ($R)"; echo; echo "The JSON line below was measured UNDER the tracer (about 8 us added per launch, ~2 000 launches per step: ms_per_step is ~15 ms above the un-instrumented run that bench.py / the driver reports); it is kept for the kernel table's context, not as the result."; echo;
  echo '```'; grep "^{\"metric\"" $OUT/bench_stdout.txt | cut -c1-3000; echo '```'; echo;
  echo "## Roofline kernels of the bench legs (rocprofv3 durations; bench.py's own HIP-event figures are in the JSON above)"; echo;
  python $T/rocpd_summary.py $DB | grep -E "^\| kernel|^\|---|k_pool<4|k_vp_|k_scan|k_gather|k_bin|k_cell|k_fill|k_conv_f32_taps|k_conv3x3_wino4_f32|k_wino4_fixup|k_wino4_wgrad|k_conv3x3_wino_f32|k_wino_wgrad|k_insert2|k_first_assign|k_gather_clean|k_conv3x3_taps|k_conv_mfma_v2|k_conv_dma_f32|k_conv3x3_wgrad_f32|k_conv1x1_wgrad_f32";
  echo; echo "## Kernels by total time, naive_conv / find-mode kernels excluded"; echo;
  python $T/rocpd_summary.py $DB | grep -v "naive_conv\|MIOpenConvUni\|Im2d2Col\|Col2Im" | head -45; } > $OUT/${R}_bench_kernel_stats.md
# 2. steady-state training step by category (marker-delimited window): fp32 (headline) and bf16
for AC in "" bf16; do
  TAG=${AC:-fp32}
  rm -rf /tmp/p_step; B=4 CL=1 AC=$AC STEPS=6 rocprofv3 --kernel-trace -d /tmp/p_step -- python $T/profile_step.py > $OUT/step_stdout_$TAG.txt 2>&1
  { echo "# Steady-state distillation step, B=4, $TAG, channels-last (6 steps between marker kernels) ($R)"; echo; echo '```'; grep "samples/s" $OUT/step_stdout_$TAG.txt; echo '```'; echo;
    TOP=30 python $T/rocpd_categories.py $(ls -S /tmp/p_step/*/*.db | head -1) 6 --top | cut -c1-180; } > $OUT/${R}_step_categories_$TAG.md
  # 2b. the same trace as a per-stream timeline of its last step (tools/rocpd_timeline.py) + the UNTRACED phase clock (tools/stream_phases.py)
  { echo "## $TAG step: streams of the last traced step (rocprofv3 --kernel-trace; the tracer adds ~8 us of host time per launch, so the traced step is host-paced: the teacher stream starts only when the host gets to it)"; echo; python $T/rocpd_timeline.py $(ls -S /tmp/p_step/*/*.db | head -1) 2; echo; echo '```'; python $T/rocpd_busy.py $(ls -S /tmp/p_step/*/*.db | head -1) 6; echo '```'; echo;
    echo "## $TAG step WITHOUT a tracer: HIP events on the stream that runs each phase (tools/stream_phases.py; ms since the step started on the GPU)"; echo; echo '```'; B=4 CL=1 AC=$AC python $T/stream_phases.py 2>&1 | tail -17; echo '```'; echo; } > $OUT/timeline_$TAG.md
done
{ echo "# Per-stream timeline of the distillation step, B = 4 ($R): main stream, weight-gradient stream, teacher stream"; echo; cat $OUT/timeline_fp32.md $OUT/timeline_bf16.md; } > $OUT/${R}_step_timeline.md
# 3. bev_pool / voxelize / dense() op level + the streaming reference points
{ echo "# bev_pool + voxelize + dense() op-level timings ($R)"; echo; echo '```'; python $T/time_bev_pool.py 2>&1 | tail -4; echo "-- voxelize, algo 0 (hash partition + LDS):"; python $T/time_voxelize.py 2>&1 | tail -16; echo "-- voxelize, algo 1 (atomic hash):"; ALGO=1 python $T/time_voxelize.py 2>&1 | tail -16; echo "-- voxelize, algo 3 (atomic hash in three launches, self-cleaning workspace; clouds of at most 256 tiles):"; ALGO=3 python $T/time_voxelize.py 2>&1 | grep -A1 "^B="; python $T/time_dense.py 2>&1 | tail -2; python $T/time_stream.py 2>&1 | tail -5; echo "-- device-scope atomics vs plain accesses (tools/atomic_rate.hip):"; hipcc --offload-arch=gfx950 -O3 $T/atomic_rate.hip -o /tmp/atomic_rate 2>/dev/null && /tmp/atomic_rate; echo '```'; } > $OUT/${R}_bevpool_voxelize_ops.md
rm -rf /tmp/p_vox; rocprofv3 --kernel-trace --stats -d /tmp/p_vox -- python $T/time_voxelize.py > /dev/null 2>&1
{ echo; echo "## voxelize kernels (algo 0; all eight configurations of tools/time_voxelize.py pooled; rocprofv3 kernel durations)"; echo; python $T/rocpd_summary.py $(ls /tmp/p_vox/*/*.db | head -1) namespace; } >> $OUT/${R}_bevpool_voxelize_ops.md
# 4. HBM traffic of the dominant kernels (separate --pmc passes, as MI355X_MICROARCH.md prescribes)
rm -rf /tmp/pmc1 /tmp/pmc2
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/pmc1 -- python $T/pmc_pool.py > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d /tmp/pmc2 -- python $T/pmc_pool.py > /dev/null 2>&1
{ echo "# HBM traffic of bev_pool.k_pool from PMC counters ($R)"; echo;
  echo "Separate passes: rocprofv3 --pmc FETCH_SIZE --kernel-trace / --pmc WRITE_SIZE --kernel-trace -- python tools/pmc_pool.py"; echo '```';
  python $T/rocpd_pmc.py $(ls /tmp/pmc1/*/*.db | head -1) k_pool | tail -1; python $T/rocpd_pmc.py $(ls /tmp/pmc2/*/*.db | head -1) k_pool | tail -1; echo '```'; } > $OUT/${R}_pmc_k_pool.md
rm -rf /tmp/pmc3 /tmp/pmc4
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/pmc3 -- python $T/time_voxelize.py > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d /tmp/pmc4 -- python $T/time_voxelize.py > /dev/null 2>&1
{ echo "# HBM traffic of the voxelize kernels (algo 0) from PMC counters ($R; all eight configurations of tools/time_voxelize.py pooled; KB)"; echo; echo '```';
  for k in k_vp_partition k_vp_bucket k_vp_flags k_scan k_vp_rows k_gather; do python $T/rocpd_pmc.py $(ls /tmp/pmc3/*/*.db | head -1) $k | tail -1; python $T/rocpd_pmc.py $(ls /tmp/pmc4/*/*.db | head -1) $k | tail -1; done; echo '```'; } > $OUT/${R}_pmc_voxelize.md
rm -rf /tmp/pmc5 /tmp/pmc6
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/pmc5 -- python $T/pmc_conv3x3.py > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d /tmp/pmc6 -- python $T/pmc_conv3x3.py > /dev/null 2>&1
{ echo "# HBM traffic of conv2d.k_conv3x3_taps (trunk 256 -> 128 @180 x 180 x 4, bf16) from PMC counters ($R; KB)"; echo; echo '```';
  python $T/rocpd_pmc.py $(ls /tmp/pmc5/*/*.db | head -1) k_conv3x3_taps | tail -1; python $T/rocpd_pmc.py $(ls /tmp/pmc6/*/*.db | head -1) k_conv3x3_taps | tail -1; echo '```'; } > $OUT/${R}_pmc_conv3x3.md
# 5. MFMA-only loop (the machine's sustained matrix rates)
{ echo "# MFMA-only loop on the MI355X box (tools/mfma_peak.hip, $R)"; echo; echo '```'; hipcc --offload-arch=gfx950 -O3 $T/mfma_peak.hip -o /tmp/mfma_peak 2>/dev/null && /tmp/mfma_peak | tee /tmp/mfma_peak.txt; echo '```'; } > $OUT/${R}_mfma_peak.md
{ echo; echo "## v_mfma_f32_32x32x16_bf16 on constant vs RANDOM operands (tools/mfma_rand.hip)"; echo; echo '```'; hipcc --offload-arch=gfx950 -O3 $T/mfma_rand.hip -o /tmp/mfma_rand 2>/dev/null && /tmp/mfma_rand; echo '```'; echo; echo "## LDS-DMA fill rate from L2-resident / HBM windows, alone and beside MFMAs (tools/dma_rate.hip)"; echo; echo '```'; hipcc --offload-arch=gfx950 -O3 $T/dma_rate.hip -o /tmp/dma_rate 2>/dev/null && /tmp/dma_rate | sed 's/ =   0.0 B\/clk\/CU@2.4//'; echo '```'; } >> $OUT/${R}_mfma_peak.md
python $T/traffic_json.py $R $(ls /tmp/pmc1/*/*.db | head -1) $(ls /tmp/pmc2/*/*.db | head -1) $(ls /tmp/pmc3/*/*.db | head -1) $(ls /tmp/pmc4/*/*.db | head -1) $(ls /tmp/pmc5/*/*.db | head -1) $(ls /tmp/pmc6/*/*.db | head -1) /tmp/mfma_peak.txt > $OUT/traffic.json 2> $OUT/traffic_json.err
# 6. sparse encoder
rm -rf /tmp/p_sp; B=4 rocprofv3 --kernel-trace --stats -d /tmp/p_sp -- python $T/time_spconv.py > $OUT/spconv_stdout.txt 2>&1
{ echo "# LiDAR sparse encoder forward, B=4 x 30k points ($R)"; echo; echo '```'; grep "encoder fwd" $OUT/spconv_stdout.txt; echo '```'; echo; python $T/rocpd_summary.py $(ls /tmp/p_sp/*/*.db | head -1) namespace | head -30; } > $OUT/${R}_spconv_encoder.md
# 6b. Winograd F(4x4,3x3): per-shape table (forward / data gradient, weight gradient) and the SQ counters of the forward kernel
{ echo "# Winograd F(4x4,3x3) fp32 kernels vs F(2x2,3x3) and the direct fp32 MFMA kernel ($R)"; echo; echo "Forward launches at the 3x3 shapes of the distillation step (tools/time_wino4.py; err = max |y - fp64| / max |fp64|; F4 with its stream-K tail):"; echo; echo '```'; python $T/time_wino4.py 2>&1 | grep "^3x3"; echo '```'; echo; echo "Weight gradient (tools/time_wino4_wgrad.py):"; echo; echo '```'; python $T/time_wino4_wgrad.py 2>&1 | grep "^wgrad"; echo '```'; echo; echo "SQ counters of k_conv3x3_wino4_f32 at 128 -> 128 @180^2 x 4 (tools/pmc_kernel.sh; sums over the chip, *_CYCLES of waves in quad-cycles, SQ_VALU_MFMA_BUSY_CYCLES in cycles):"; echo; echo '```'; bash $T/pmc_kernel.sh k_conv3x3_wino4_f32 python $T/pmc_wino4.py 2>&1 | grep " n=" | sed "s/.*k_conv3x3_wino4_f32[^ ]* [^ ]* [^ ]* [a-z]* */k_conv3x3_wino4_f32  /" | awk '{printf "%-22s %-26s %s %s\n", $1, $2, $3, $4}'; echo '```'; } > $OUT/${R}_conv_f32_wino4.md
# 6c. fp32 1x1 family: persistent stream-K / per-tile kernels vs the grid-per-tile kernel, per plain shape of one step; SQ counters
{ echo "# fp32 1x1 convolutions: ud_conv1x1p_nhwc_f32 (persistent stream-K from 12 slices, per-tile with register epilogue below) vs the grid-per-tile kernel ($R)"; echo; echo "tools/time_1x1p.py (plain launches of one distillation step; 'new' = the launcher's own schedule, 'no-SK' = without the workspace; err vs an fp64 matmul with bias + residual + ReLU + BatchNorm sums):"; echo; echo '```'; python $T/time_1x1p.py 2>&1 | grep -E "^ +[0-9P]|per step"; echo '```'; echo; echo "SQ counters of k_conv1x1p_f32 at 16 896 x 1024 -> 256 with its stream-K tail (tools/pmc_kernel.sh k_conv1x1p python tools/pmc_1x1p.py 16896 1024 256 1):"; echo; echo '```'; bash $T/pmc_kernel.sh k_conv1x1p python $T/pmc_1x1p.py 16896 1024 256 1 2>&1 | grep " n=" | grep "k_conv1x1p_f32" | sed "s/.*k_conv1x1p_f32[^ ]* [^ ]* [^ ]* [a-z]* */k_conv1x1p_f32  /" | awk '{printf "%-18s %-26s %s %s\n", $1, $2, $3, $4}'; echo '```'; } > $OUT/${R}_conv_f32_1x1.md
# 7. fp32 convolutions: ours vs library (forward / data gradient, and the weight gradients of one step)
{ echo "# fp32 convolutions: hand-written fp32 MFMA kernels (direct and Winograd: the launcher's routing) vs MIOpen ($R)"; echo; echo '```'; python $T/time_conv2d_f32.py 2>&1 | tail -13; echo; echo "-- weight gradients of one distillation step (tools/time_f32_wgrad.py; 3x3: Winograd form, last column = the direct kernel):"; python $T/time_f32_wgrad.py 2>&1 | tail -22; echo; echo "-- plain 1x1 launches of one step (tools/time_f32_1x1.py):"; python $T/time_f32_1x1.py 2>&1 | grep -E "kind|line|total"; echo; echo "-- frozen ResNet stem, 24 x 256 x 704 (tools/time_stem.py):"; python $T/time_stem.py 2>&1 | grep "us "; echo '```'; } > $OUT/${R}_conv_f32.md
# 8. host lead: how far the enqueueing threads run ahead of the GPU at the phase boundaries of a step (fp32 and bf16)
{ echo "# Host lead over the GPU inside a training step (tools/host_lead.py, $R)"; echo; echo "Host timestamps and HIP events at the same points of Trainer.step; lead = GPU time - host time at that point (ms since the loop start). A positive lead at every boundary = the step is GPU-bound, the host enqueue cost is hidden."; echo; echo "fp32:"; echo '```'; python $T/host_lead.py 2>&1 | tail -11; echo '```'; echo; echo "bf16 autocast:"; echo '```'; AC=bf16 python $T/host_lead.py 2>&1 | tail -11; echo '```'; } > $OUT/${R}_host_lead.md
# 8b. the mapped (strided / transposed / im2col / dgrad-class) 1x1 launches of one step through their routing, and SQ counters of the spconv kernel
{ echo "# Mapped 1x1 launches of one distillation step (tools/time_f32_1x1.py MAPPED=1, $R; the total line counts the plain launches of profiles/${R}_conv_f32.md too)"; echo; echo '```'; MAPPED=1 python $T/time_f32_1x1.py 2>&1 | grep -E "kind|mapped|total"; echo '```'; echo; echo "SQ counters of k_conv_dma_f32 (sparse 128 -> 128 layers, B=4 encoder pass; tools/pmc_kernel.sh):"; echo; echo '```'; B=4 bash $T/pmc_kernel.sh k_conv_dma_f32 python $T/time_spconv.py 2>&1 | grep " n=" | grep "128, 128" | awk '{printf "k_conv_dma_f32<128,128>  %-26s %s %s\n", $(NF-4), $(NF-3), $(NF-2)}'; echo '```'; } > $OUT/${R}_conv_f32_mapped.md
ls -la $OUT
