#!/bin/bash
# counters of the all-pairs kernels at n = 128, log2m 20, K 31 (scripts/bench_gram_one.py), one rocprofv3 pass per set:
#   pmc_gram.sh OUTNAME   -- environment (DD_GRAM_DIAG2, DD_GRAM_XCD, GRAM_ZEROS ...) passes through
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/${1:-gram_pmc}; mkdir -p $OUT; : > $OUT/pmc.txt
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA" \
           "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/raw$i -o x -- python3 scripts/bench_gram_one.py 128 > /dev/null 2>&1
  python3 scripts/pmc_summary.py $(find $OUT/raw$i -name "*counter_collection.csv" | head -1) | grep -A10 "^gram_kernel" >> $OUT/pmc.txt
  rm -rf $OUT/raw$i
done
cat $OUT/pmc.txt
