#!/bin/bash
# The seven fuzzers on the current build, one sitting (numbers for DESIGN.md section 6): fuzz_all.sh [SCALE] [SEED_BASE]   (SCALE 1 = ~10 min)
S=${1:-1}; B=${2:-7050}; mkdir -p gpurun_out; OUT=gpurun_out/fuzz_all.txt; : > $OUT
run() { echo "== $*" | tee -a $OUT; timeout 1500 python "$@" 2>&1 | tail -2 | tee -a $OUT; }
run scripts/fuzz_parity.py $((6000*S)) $((B+1))
run scripts/fuzz_buckets.py $((4000*S)) $((B+2))
run scripts/fuzz_inflate.py $((500*S)) $((B+3))
run scripts/fuzz_damage.py $((2000*S)) $((B+4))
run scripts/fuzz_fastq.py $((5000*S)) $((B+5))
run scripts/fuzz_k2.py $((150*S)) $((B+6))
run scripts/fuzz_cli.py $((1000*S)) $((B+7))
