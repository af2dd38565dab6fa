#!/bin/bash
# PMC counters of the K2 all-pairs kernels (dd_gram.hip) on scripts/bench_k2.py: usage pmc_k2.sh OUTNAME LOG2M
# (SQ set, FETCH_SIZE and WRITE_SIZE in separate passes, as MI355X_MICROARCH.md prescribes)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/${1:-k2_pmc}; P=${2:-20}; mkdir -p $OUT; : > $OUT/pmc_p$P.txt
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA" \
           "GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
  i=$((i+1))
  timeout 400 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/raw$i -o x -- python3 scripts/bench_k2.py $P pairwise > /dev/null 2>&1
  python3 scripts/pmc_summary.py $(find $OUT/raw$i -name "*counter_collection.csv" | head -1) | grep -A12 "^gram\|^pairwise" >> $OUT/pmc_p$P.txt
  rm -rf $OUT/raw$i
done
cat $OUT/pmc_p$P.txt
