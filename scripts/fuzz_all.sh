#!/bin/bash
# The six fuzzers on the current build, one sitting (numbers for DESIGN.md section 6): fuzz_all.sh [SCALE]   (SCALE 1 = ~10 min)
S=${1:-1}; mkdir -p gpurun_out; OUT=gpurun_out/fuzz_all.txt; : > $OUT
run() { echo "== $*" | tee -a $OUT; timeout 1500 python "$@" 2>&1 | tail -2 | tee -a $OUT; }
run scripts/fuzz_parity.py $((6000*S)) 7051
run scripts/fuzz_buckets.py $((4000*S)) 7052
run scripts/fuzz_inflate.py $((500*S)) 7053
run scripts/fuzz_damage.py $((2000*S)) 7054
run scripts/fuzz_fastq.py $((5000*S)) 7055
run scripts/fuzz_k2.py $((150*S)) 7056
