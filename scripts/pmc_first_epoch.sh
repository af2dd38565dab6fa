#!/bin/bash
# SQ / LDS counters of the first-epoch scatter and its replay, one k class alone on the chip (64 x 5 Mbp, k 10..16, log2m $1)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
P=${1:-20}; OUT=gpurun_out/pmc_first_p$P; mkdir -p $OUT; : > $OUT/pmc.txt
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" \
           "GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD" ; do
  i=$((i+1))
  timeout 400 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/raw$i -o x -- python3 scripts/quick_bench.py 64 5e6 10 16 $P > /dev/null 2>&1
  python3 scripts/pmc_summary.py $(find $OUT/raw$i -name "*counter_collection.csv" | head -1) | grep -A9 "^scatter_kernel\|^replay_kernel" >> $OUT/pmc.txt
  rm -rf $OUT/raw$i
done
cat $OUT/pmc.txt
