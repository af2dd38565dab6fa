run() { P=$1; shift; echo "== log2m $P $*"; env "$@" python scripts/quick_bench.py 10 50e6 4 40 $P | sed -n 3p; }
for e in 4 16 32; do run 18 DD_BUCKET_E0=$e; done
run 18 DD_BUCKET_EMAX=64
run 18 DD_BUCKET_EMAX=128
for e in 8 16 32; do run 20 DD_BUCKET_E0=$e; done
run 20 DD_BUCKET_EMAX=64
run 20 DD_BUCKET_EMAX=96
run 20 DD_BUCKET_EMAX=128
run 20 DD_BUCKET_LOGG=4
run 20 DD_BUCKET_FBITS=8
run 18 DD_BUCKET_FBITS=8
