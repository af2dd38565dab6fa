#!/bin/bash
# Round 5, VERDICT r04 #3a: row-group epochs.  A single-epoch call at log2m >= 17 (64 x 5 Mbp at -r 20: every update a record,
# 87 GB through HBM per step) runs scatter -> replay per group of rows whose record areas sum to DD_ROW_GROUP_MB, the areas of a
# stream's successive groups being the same ring of slots -- does the 256 MiB memory-side cache hold the records between the
# write and the read?  A/B on one box, kernels only (scripts/quick_bench.py).  Writes gpurun_out/row_groups.txt
mkdir -p gpurun_out
OUT=gpurun_out/row_groups.txt
: > $OUT
if [ "$1" != "notest" ]; then
  timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "bucket or row_groups" 2>&1 | tail -5 | tee -a $OUT
fi
for rep in 1 2; do
  for mb in 0 90 180 360 720 1440; do
    for tpj in 1 2; do
      if [ $mb = 0 ] && [ $tpj = 2 ]; then continue; fi
      for cfg in "64 5e6 10 40 20" "64 5e6 4 40 20" "64 5e6 10 40 18"; do
        echo "== DD_ROW_GROUP_MB=$mb TPJ=$tpj  quick_bench $cfg" | tee -a $OUT
        DD_ROW_GROUP_MB=$mb DD_ROW_GROUP_TPJ=$tpj timeout 300 python scripts/quick_bench.py $cfg 2>&1 | grep -E "iter [12]|Error|fault" | tee -a $OUT
      done
    done
  done
done
