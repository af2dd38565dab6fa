#!/bin/bash
{
for E in 32 64 96 128; do echo "== 10 x 50 p=20 E0=$E"; DD_BUCKET_E0=$E timeout 300 python scripts/quick_bench.py 10 50e6 4 40 20 2>&1 | grep "iter 2"; done
for E in 16 32 64; do echo "== 10 x 50 p=19 E0=$E"; DD_BUCKET_E0=$E timeout 300 python scripts/quick_bench.py 10 50e6 4 40 19 2>&1 | grep "iter 2"; done
for E in 8 16 32 48; do echo "== 10 x 50 p=18 E0=$E"; DD_BUCKET_E0=$E timeout 300 python scripts/quick_bench.py 10 50e6 4 40 18 2>&1 | grep "iter 2"; done
for E in 4 8 16 24; do echo "== 10 x 50 p=17 E0=$E"; DD_BUCKET_E0=$E timeout 300 python scripts/quick_bench.py 10 50e6 4 40 17 2>&1 | grep "iter 2"; done
for E in 32 64 77; do echo "== 64 x 5 p=20 E0=$E"; DD_BUCKET_E0=$E timeout 300 python scripts/quick_bench.py 64 5e6 4 40 20 2>&1 | grep "iter 2"; done
for E in 8 16 32 77; do echo "== 64 x 5 p=18 E0=$E"; DD_BUCKET_E0=$E timeout 300 python scripts/quick_bench.py 64 5e6 4 40 18 2>&1 | grep "iter 2"; done
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/exp_e0b.txt
