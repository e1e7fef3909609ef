#!/bin/bash
# SQ counter passes over any tool script: gpurun -- 'bash tools/pmc_kernel.sh <kernel-name-filter> python tools/<script>.py'
# Output: gpurun_out/pmc_<filter>.txt (per-kernel averages; SQ_* are summed over the chip, *_CYCLES of waves in quad-cycles).
F=$1; shift
cd /tmp && export TMPDIR=/tmp
T=$GRAFT_REPO_ROOT/tools; O=$GRAFT_REPO_ROOT/gpurun_out/pmc_$F.txt; : > $O
i=0
for SET in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_LDS_UNALIGNED_STALL" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_WAVES"; do
  i=$((i+1)); rm -rf /tmp/pk$i
  timeout 600 rocprofv3 --pmc $SET --kernel-trace -d /tmp/pk$i -- "$@" > /tmp/pk$i.log 2>&1 || { echo "pass $i failed: $(tail -n 3 /tmp/pk$i.log)" >> $O; continue; }
  python $T/rocpd_pmc.py $(ls /tmp/pk$i/*/*.db | head -1) $F | grep -v "^columns" >> $O 2>&1
done
cat $O
