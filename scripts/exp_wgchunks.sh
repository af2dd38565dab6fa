#!/bin/bash
# Round 4, first epoch of the log2m >= 17 path: binned tiles of tokens (DD_FIRST_WG=3, default) against the workgroup's
# sorted 16 384-record chunks (2) and round 3's per-wave 1024-record chunks (0), alternating on one box.
# Writes gpurun_out/wgchunks.txt
mkdir -p gpurun_out
OUT=gpurun_out/wgchunks.txt
: > $OUT
if [ "$1" != "notest" ]; then
  timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "bucket or realistic or inputs_without" 2>&1 | tail -5 | tee -a $OUT
fi
for rep in 1 2; do
  for wg in 0 2 3; do
    for cfg in "64 5e6 10 40 20" "10 50e6 4 40 20" "64 5e6 10 40 18" "10 50e6 4 40 18"; do
      echo "== DD_FIRST_WG=$wg  quick_bench $cfg" | tee -a $OUT
      DD_FIRST_WG=$wg timeout 300 python scripts/quick_bench.py $cfg 2>&1 | grep -E "iter [12]|Error|fault" | tee -a $OUT
    done
  done
done
