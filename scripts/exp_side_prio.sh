# stream priorities of the k-class pipelines (DD_SIDE_PRIO): 64 x 5 Mbp and 10 x 50 Mbp at log2m 20
cd $GRAFT_REPO_ROOT
for pr in none 012 021 102 120 201 210 000 222; do
  if [ $pr = none ]; then unset DD_SIDE_PRIO; else export DD_SIDE_PRIO=$pr; fi
  a=$(python3 scripts/quick_bench.py 64 5e6 10 40 20 | grep "iter" | awk '{print $4}' | sort -n | head -1)
  b=$(python3 scripts/quick_bench.py 10 50e6 4 40 20 | grep "iter" | awk '{print $4}' | sort -n | head -1)
  echo "prio $pr: 64x5Mbp $a ms   10x50Mbp $b ms"
done
