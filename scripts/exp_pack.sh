#!/bin/bash
# Round 5: the first epoch's bins packed to 3 bytes per record through an LDS ring (DD_FIRST_WG=4, scatter_first_pack_kernel +
# replay FORM 4) against round 4's 4-byte bins (DD_FIRST_WG=3), alternating on one box.  Writes gpurun_out/pack.txt
mkdir -p gpurun_out
OUT=gpurun_out/pack.txt
: > $OUT
if [ "$1" != "notest" ]; then
  timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "bucket or row_groups or packed or realistic or inputs_without or register_sizes" 2>&1 | tail -5 | tee -a $OUT
fi
for rep in 1 2; do
  for wg in 3 4; do
    for cfg in "64 5e6 10 40 20" "64 5e6 4 40 20" "10 50e6 4 40 20" "64 5e6 10 40 18" "10 50e6 4 40 18" "10 50e6 4 40 17" "4 300e6 49 64 20"; do
      echo "== DD_FIRST_WG=$wg  quick_bench $cfg" | tee -a $OUT
      DD_FIRST_WG=$wg timeout 300 python scripts/quick_bench.py $cfg 2>&1 | grep -E "iter [12]|Error|fault" | tee -a $OUT
    done
  done
done
