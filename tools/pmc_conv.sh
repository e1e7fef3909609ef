#!/bin/bash
# SQ counter passes over one conv3x3 bf16 shape (gpurun -- 'bash tools/pmc_conv.sh'); output: gpurun_out/pmc_conv.txt
cd /tmp && export TMPDIR=/tmp
T=$GRAFT_REPO_ROOT/tools; O=$GRAFT_REPO_ROOT/gpurun_out/pmc_conv.txt; : > $O
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z_0-9]*\|TCP_[A-Z_0-9]*\|TA_[A-Z_0-9]*" | sort -u | tr '\n' ' ' > $GRAFT_REPO_ROOT/gpurun_out/pmc_names.txt
i=0
for SET in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_LDS_UNALIGNED_STALL" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_INST_CYCLES_VMEM SQ_WAVES SQ_INSTS_VALU_MFMA_MOPS_BF16" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES SQ_INST_CYCLES_VMEM"; do
  i=$((i+1)); rm -rf /tmp/pc$i
  timeout 300 rocprofv3 --pmc $SET --kernel-trace -d /tmp/pc$i -- python $T/pmc_conv.py > /tmp/pc$i.log 2>&1 || { echo "pass $i failed: $(tail -n 3 /tmp/pc$i.log)" >> $O; continue; }
  python $T/rocpd_pmc.py $(ls /tmp/pc$i/*/*.db | head -1) k_conv3x3 >> $O 2>&1
done
cat $O
