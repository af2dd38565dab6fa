#!/bin/bash
# per-k cost of the log2m >= 18 path: one k at a time over 10 x 50 Mbp (last iteration's wall time)
P=${1:-20}
mkdir -p gpurun_out
for k in 4 9 10 11 12 13 14 16 17 24 32 33 40; do
  echo -n "k=$k p=$P: "; timeout 120 python scripts/quick_bench.py 10 50e6 $k $k $P | grep "iter 2" 
done 2>&1 | tee gpurun_out/per_k_p$P.txt
