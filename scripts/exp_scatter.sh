# timing experiments on the log2m >= 18 path (results of the DEBUG runs are wrong on purpose)
P=${1:-20}
run() { echo "== $*"; env "$@" python scripts/quick_bench.py 10 50e6 4 40 $P | sed -n 3p; }
run X=0
run DD_SCATTER_DEBUG=1
run DD_BUCKET_EMAX=64
run DD_BUCKET_LOGG=3
run DD_BUCKET_LOGG=5
run DD_NO_XCD_AFFINITY=1
echo "== 1 genome x 500 Mbp"; python scripts/quick_bench.py 1 500e6 4 40 $P | sed -n 3p
echo "== 2 genomes x 50 Mbp"; python scripts/quick_bench.py 2 50e6 4 40 $P | sed -n 3p
echo "== 64 genomes x 5 Mbp k 2..32"; python scripts/quick_bench.py 64 5e6 2 32 $P | sed -n 3p
