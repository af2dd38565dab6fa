run() { P=$1; shift; echo "== log2m $P $*"; env "$@" python scripts/quick_bench.py 10 50e6 4 40 $P | sed -n 3p; }
run 18 DD_BUCKET_LOGG=0
run 18 DD_BUCKET_LOGG=1
run 19 DD_BUCKET_LOGG=1
run 19 DD_BUCKET_LOGG=2
run 20 DD_BUCKET_LOGG=2
run 20 DD_BUCKET_LOGG=3
