run() { P=$1; shift; echo "== log2m $P $*"; env "$@" python scripts/quick_bench.py 10 50e6 4 40 $P | sed -n 3p; }
for e in 2 4 8 16 32; do run 18 DD_BUCKET_E0=$e; done
for e in 8 16 32 64; do run 20 DD_BUCKET_E0=$e; done
run 20 DD_BUCKET_EMAX=128
run 20 DD_BUCKET_GB=32
run 18 DD_BUCKET_LOGG=3
run 18 DD_BUCKET_LOGG=1
run 19 X=1
