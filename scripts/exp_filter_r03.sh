# what a coarser group filter costs by itself (one k per job): the price of putting 2 (32 KiB filters) or 4 (16 KiB)
# ks into one scatter workgroup at two workgroups per CU, before any gain from the shared window push
cd $GRAFT_REPO_ROOT
run() { P=$1; shift; t=$(env "$@" python3 scripts/quick_bench.py 10 50e6 4 40 $P | grep iter | awk '{print $4}' | sort -n | head -1); echo "10 x 50 Mbp log2m $P $*: $t ms"; }
run 20 DD_BUCKET_LOGG=3
run 20 DD_BUCKET_LOGG=4
run 20 DD_BUCKET_LOGG=5
run 18 DD_BUCKET_LOGG=1
run 18 DD_BUCKET_LOGG=2
run 18 DD_BUCKET_LOGG=3
