cd $GRAFT_REPO_ROOT
for sg in 0 1; do
  if [ $sg = 0 ]; then unset DD_BUCKET_STAGGER; else export DD_BUCKET_STAGGER=1; fi
  for p in 18 20; do
  a=$(python3 scripts/quick_bench.py 64 5e6 10 40 $p | grep "iter" | awk '{print $4}' | sort -n | head -1)
  b=$(python3 scripts/quick_bench.py 10 50e6 4 40 $p | grep "iter" | awk '{print $4}' | sort -n | head -1)
  echo "stagger $sg log2m $p: 64x5Mbp $a ms   10x50Mbp $b ms"
  done
done
