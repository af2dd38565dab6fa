#!/bin/bash
# dense per-row record stream: parity of the bucket path, then timings at log2m 18 / 19 / 20
mkdir -p gpurun_out
{
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "bucket or batched or 18 or 20 or global" 2>&1 | tail -4
for P in 18 19 20; do
  echo "== log2m $P"; timeout 200 python scripts/quick_bench.py 10 50e6 4 40 $P | grep "iter 2"
done
export DD_BUCKET_SLOTS=8192
for E in 48 64 96 128; do echo "== slots 8192 E0 $E log2m 20";  DD_BUCKET_E0=$E timeout 200 python scripts/quick_bench.py 10 50e6 4 40 20 | grep "iter 2"; done
for E in 16 32 48 64; do echo "== slots 8192 E0 $E log2m 19";  DD_BUCKET_E0=$E timeout 200 python scripts/quick_bench.py 10 50e6 4 40 19 | grep "iter 2"; done
for E in 8 16 24 32; do echo "== slots 8192 E0 $E log2m 18";  DD_BUCKET_E0=$E timeout 200 python scripts/quick_bench.py 10 50e6 4 40 18 | grep "iter 2"; done
for S in 6144 12288; do echo "== slots $S E0 64 log2m 20";  DD_BUCKET_SLOTS=$S DD_BUCKET_E0=64 timeout 200 python scripts/quick_bench.py 10 50e6 4 40 20 | grep "iter 2"; done
timeout 300 python scripts/fuzz_buckets.py 300 7 2>&1 | tail -2
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/exp_dense.txt
