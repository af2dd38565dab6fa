#!/bin/bash
{
for P in 17 18; do
 for E in 4 8 16; do for S in 4096 8192; do
  echo -n "p=$P E0=$E slots=$S: "; DD_BUCKET_E0=$E DD_BUCKET_SLOTS=$S timeout 200 python scripts/quick_bench.py 10 50e6 4 40 $P | grep "iter 2"
 done; done
done
for U in 2 4 8; do echo -n "p=17 unit=$U: "; DD_BUCKET_UNIT=$U timeout 200 python scripts/quick_bench.py 10 50e6 4 40 17 | grep "iter 2"; done
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/exp_tune.txt
