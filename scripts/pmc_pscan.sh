#!/bin/bash
# PMC counters of the progressive kernels on scripts/bench_k2.py: usage pmc_pscan.sh OUTNAME LOG2M
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/${1:-pscan_pmc}; P=${2:-20}; mkdir -p $OUT; : > $OUT/pmc_p$P.txt
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" \
           "GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD" ; do
  i=$((i+1))
  timeout 400 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/raw$i -o x -- python3 scripts/bench_k2.py $P > /dev/null 2>&1
  python3 scripts/pmc_summary.py $(find $OUT/raw$i -name "*counter_collection.csv" | head -1) | grep -A9 "^pscan\|^progressive" >> $OUT/pmc_p$P.txt
  rm -rf $OUT/raw$i
done
cat $OUT/pmc_p$P.txt
