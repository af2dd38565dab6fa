# two ks per filtered scatter job (shared token extraction and window push) with filters small enough for two workgroups per CU
cd $GRAFT_REPO_ROOT
python3 -m pytest tests/test_gpu_parity.py -q -x -k "bucket_mode_knobs" 2>&1 | tail -2
run() { P=$1; shift; t=$(env "$@" python3 scripts/quick_bench.py 10 50e6 4 40 $P | grep iter | awk '{print $4}' | sort -n | head -1); echo "10 x 50 Mbp log2m $P $*: $t ms"; }
run 20 DD_BUCKET_LOGG=3
run 20 DD_BUCKET_LOGG=5
run 20 DD_BUCKET_NK=2 DD_BUCKET_LOGG=5
run 20 DD_BUCKET_NK=2 DD_BUCKET_LOGG=4
run 20 DD_BUCKET_NK=2 DD_BUCKET_LOGG=3
run 18 DD_BUCKET_LOGG=1
run 18 DD_BUCKET_NK=2 DD_BUCKET_LOGG=3
run 18 DD_BUCKET_NK=2 DD_BUCKET_LOGG=2
run 18 DD_BUCKET_NK=2 DD_BUCKET_LOGG=1
