#!/bin/bash
# k classes of the log2m >= 18 path in isolation
P=${1:-20}
for r in "4 9" "10 16" "12 16" "10 11" "17 32" "33 40" "10 40" "4 40"; do
  set -- $r
  echo -n "k $1..$2 p=$P: "; timeout 120 python scripts/quick_bench.py 10 50e6 $1 $2 $P 2>&1 | grep "iter 2"
done 2>&1 | grep -v amdgpu.ids | tee gpurun_out/classes_p$P.txt
