#!/bin/bash
# Regenerate the rocprofv3 evidence on the GPU box (run through gpurun); summaries land in
# gpurun_out/profiles_rNN/ and are then copied to profiles/ (tracked).
#   gpurun --timeout 1500 -- 'bash tools/make_profiles.sh r01'
set -u
R=${1:-r01}
OUT=$GRAFT_REPO_ROOT/gpurun_out/profiles_$R
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
T=$GRAFT_REPO_ROOT/tools
# 1. the bench command itself: kernel trace + stats
rm -rf /tmp/p_bench; rocprofv3 --kernel-trace --stats -d /tmp/p_bench -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline > $OUT/bench_stdout.txt 2>&1
DB=$(ls /tmp/p_bench/*/*.db | head -1)
{ echo "# rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline  ($R)"; echo;
  echo '```'; grep "^{\"metric\"" $OUT/bench_stdout.txt; echo '```'; echo; echo "## Kernels by total time (whole run incl. warm-up, MIOpen find and the roofline leg)"; echo;
  python $T/rocpd_summary.py $DB | head -60;
  echo; echo "## Roofline kernel (bench.py roofline leg: 2 warm-up + 10 timed launches of bev_pool forward, cache scrubbed)"; echo;
  python $T/rocpd_summary.py $DB | grep -E "^\| kernel|^\|---|k_pool<4"; } > $OUT/${R}_bench_kernel_stats.md
# 2. steady-state training step by category (marker-delimited window)
rm -rf /tmp/p_step; B=4 CL=1 AC=bf16 STEPS=6 rocprofv3 --kernel-trace -d /tmp/p_step -- python $T/profile_step.py > $OUT/step_stdout.txt 2>&1
{ echo "# Steady-state distillation step, B=4, bf16 + channels-last (6 steps between marker kernels) ($R)"; echo; echo '```'; grep "samples/s" $OUT/step_stdout.txt; echo '```'; echo;
  python $T/rocpd_categories.py $(ls /tmp/p_step/*/*.db | head -1) 6 --top; } > $OUT/${R}_step_categories.md
# 3. bev_pool / voxelize op level
{ echo "# bev_pool + voxelize op-level timings ($R)"; echo; echo '```'; python $T/time_bev_pool.py 2>&1 | tail -4; python $T/exp_pool.py 2>&1 | tail -4; python $T/time_voxelize.py 2>&1 | tail -6; python $T/exp_stream.py 2>&1 | tail -5; echo '```'; } > $OUT/${R}_bevpool_voxelize_ops.md
rm -rf /tmp/p_vox; rocprofv3 --kernel-trace --stats -d /tmp/p_vox -- python $T/time_voxelize.py > /dev/null 2>&1
{ echo; echo "## voxelize kernels (all six configurations of tools/time_voxelize.py pooled)"; echo; python $T/rocpd_summary.py $(ls /tmp/p_vox/*/*.db | head -1) namespace; } >> $OUT/${R}_bevpool_voxelize_ops.md
# 4. HBM traffic of the dominant kernel (separate --pmc passes, as MI355X_MICROARCH.md prescribes)
rm -rf /tmp/pmc1 /tmp/pmc2
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/pmc1 -- python $T/pmc_pool.py > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d /tmp/pmc2 -- python $T/pmc_pool.py > /dev/null 2>&1
{ echo "# HBM traffic of bev_pool.k_pool from PMC counters ($R)"; echo;
  echo "Separate passes: rocprofv3 --pmc FETCH_SIZE --kernel-trace / --pmc WRITE_SIZE --kernel-trace -- python tools/pmc_pool.py"; echo '```';
  python $T/rocpd_pmc.py $(ls /tmp/pmc1/*/*.db | head -1) k_pool | tail -1; python $T/rocpd_pmc.py $(ls /tmp/pmc2/*/*.db | head -1) k_pool | tail -1; echo '```'; } > $OUT/${R}_pmc_k_pool.md
# 5. sparse encoder
rm -rf /tmp/p_sp; B=4 rocprofv3 --kernel-trace --stats -d /tmp/p_sp -- python $T/time_spconv.py > $OUT/spconv_stdout.txt 2>&1
{ echo "# LiDAR sparse encoder forward, B=4 x 30k points ($R)"; echo; echo '```'; grep "encoder fwd" $OUT/spconv_stdout.txt; echo '```'; echo; python $T/rocpd_summary.py $(ls /tmp/p_sp/*/*.db | head -1) namespace | head -30; } > $OUT/${R}_spconv_encoder.md
ls -la $OUT
