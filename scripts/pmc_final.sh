#!/bin/bash
# counters of the final log2m 20 step: global atomics, then instruction counts, per kernel (separate passes)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/${1:-r02_pmc_final}; mkdir -p $OUT; : > $OUT/pmc.txt
i=0
for set in "TCC_ATOMIC_sum TCC_EA0_ATOMIC_sum TCC_REQ_sum" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY"; do
  i=$((i+1))
  timeout 400 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/raw$i -o x -- python3 scripts/quick_bench.py 10 50e6 4 40 20 > /dev/null 2>&1
  python3 scripts/pmc_any.py $(find $OUT/raw$i -name "*counter_collection.csv" | head -1) >> $OUT/pmc.txt
  rm -rf $OUT/raw$i
done
cat $OUT/pmc.txt
