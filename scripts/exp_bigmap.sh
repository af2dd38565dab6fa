#!/bin/bash
mkdir -p gpurun_out
{
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -x -q 2>&1 | tail -4
timeout 300 python scripts/fuzz_buckets.py 600 21 2>&1 | tail -2
for P in 19 20; do
  echo "== log2m $P"; timeout 200 python scripts/quick_bench.py 10 50e6 4 40 $P | grep "iter 2"
  echo "== log2m $P no bigmap"; DD_NO_BIGMAP=1 timeout 200 python scripts/quick_bench.py 10 50e6 4 40 $P | grep "iter 2"
done
echo "== k 10..11 p 20"; timeout 200 python scripts/quick_bench.py 10 50e6 10 11 20 | grep "iter 2"
echo "== k 10..16 p 20"; timeout 200 python scripts/quick_bench.py 10 50e6 10 16 20 | grep "iter 2"
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/exp_bigmap.txt
