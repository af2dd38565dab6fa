#!/bin/bash
# L2 / fabric-side counters of the first-epoch scatter and its replay, one k class alone (64 x 5 Mbp, k 10..16, log2m $1)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
P=${1:-20}; OUT=gpurun_out/pmc_first_l2_p$P; mkdir -p $OUT; : > $OUT/pmc.txt
i=0
for set in "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_BUSY_sum" \
           "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TA_BUSY_sum" \
           "GRBM_GUI_ACTIVE TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum" ; do
  i=$((i+1))
  timeout 400 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/raw$i -o x -- python3 scripts/quick_bench.py 64 5e6 10 16 $P > /dev/null 2>&1
  python3 scripts/pmc_summary.py $(find $OUT/raw$i -name "*counter_collection.csv" | head -1) | grep -A5 "^scatter_kernel\|^replay_kernel" >> $OUT/pmc.txt
  rm -rf $OUT/raw$i
done
cat $OUT/pmc.txt
