# does bringing the GPU context up on a thread beside the digests help a cold `dandd tree`?  (same box, alternating)
cd $GRAFT_REPO_ROOT
for r in 14 20; do for i in 1 2 3 4; do for mode in prewarm none; do
  if [ $mode = none ]; then export DANDD_NO_PREWARM=1; else unset DANDD_NO_PREWARM; fi
  t=$(python3 scripts/e2e_cli.py 10 50 --registers $r 2>/dev/null | grep workload | python3 -c "import sys,json; print(json.loads(sys.stdin.read())['seconds']['tree'])")
  echo "log2m $r $mode tree $t"
done; done; done
