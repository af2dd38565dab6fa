# timing experiments on the log2m >= 18 path (results of the DEBUG runs are wrong on purpose)
P=${1:-20}
run() { echo "== $*"; env "$@" python scripts/quick_bench.py 10 50e6 4 40 $P | sed -n 3p; }
run X=0
run DD_SCATTER_DEBUG=1
run DD_SCATTER_DEBUG=2
run DD_BUCKET_GB=2
run DD_BUCKET_GB=4 
run DD_BUCKET_EMAX=32
run DD_BUCKET_LOGG=4
run DD_BUCKET_LOGG=6
run DD_NO_XCD_AFFINITY=1
run DD_NO_BUCKETS=1
echo "== 1 genome x 500 Mbp"; python scripts/quick_bench.py 1 500e6 4 40 $P | sed -n 3p
echo "== 2 genomes x 50 Mbp"; python scripts/quick_bench.py 2 50e6 4 40 $P | sed -n 3p
