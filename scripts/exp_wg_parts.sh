#!/bin/bash
# timing-only builds of scatter_first_wg_kernel (build/libdandd_exp$e.so, -DDD_EXP=e): where its time goes
# 0 = product, 1 = no counting atomic, 2 = no placement, 3 = placement without the returning atomic (coalesced), 4 = no workgroup barrier, 5 = hash only
for e in 0 1 2 3 4 5; do
  for ks in "10 16" "17 32"; do
    if [ $e = 0 ]; then unset DANDD_LIB; else export DANDD_LIB=build/libdandd_exp$e.so; fi
    echo "== exp $e k $ks"
    bash scripts/prof_any.sh wgparts_$e 64 5e6 $ks 20 2>&1 | grep -E "iter 2|scatter|replay"
  done
done
