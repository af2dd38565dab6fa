#!/bin/bash
{
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -2
timeout 300 python scripts/fuzz_buckets.py 1500 404 2>&1 | tail -1
for cfg in "" "DD_NO_PRESORT=1"; do
  for P in 17 18 19 20; do echo "== 10 x 50 p=$P $cfg"; env $cfg timeout 300 python scripts/quick_bench.py 10 50e6 4 40 $P 2>&1 | grep "iter 2"; done
  for P in 18 20; do echo "== 64 x 5 p=$P $cfg"; env $cfg timeout 300 python scripts/quick_bench.py 64 5e6 4 40 $P 2>&1 | grep "iter 2"; done
done
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/exp_presort.txt
