#!/bin/bash
# round 4: window classes as 32-bit words in sweep_kernel (log2m <= 16) -- build/libdandd_w4.so (128-bit class), w7 (all three)
for rep in 1 2; do
for lib in "" build/libdandd_w4.so build/libdandd_w7.so; do
  for cfg in "4 300e6 49 64 14" "4 300e6 49 64 16" "4 300e6 33 48 16" "4 300e6 17 32 16" "4 300e6 4 64 16" "10 50e6 4 40 14"; do
    echo "== lib=$lib $cfg"; DANDD_LIB=$lib python scripts/quick_bench.py $cfg | grep "iter 2"
  done
done
done
echo "== scatter 128-bit class at log2m 20 (product build: Windows<7>)"; python scripts/quick_bench.py 4 300e6 49 64 20 | grep "iter 2"
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "sweep_parity or bucket_mode_knobs" 2>&1 | tail -2
DANDD_LIB=build/libdandd_w7.so timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "sweep_parity" 2>&1 | tail -2
