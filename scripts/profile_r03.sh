#!/bin/bash
# Round-3 counter evidence for K1 at log2m 14 and 20 (bench.py's cfg 2 call: 10 x 50 Mbp, k 4-40), one rocprofv3 pass
# per counter set as MI355X_MICROARCH.md prescribes (FETCH_SIZE and WRITE_SIZE never share a pass; --pmc runs carry
# --kernel-trace only).  Raw CSVs stay in gpurun_out/; scripts/make_counters_json.py turns them into
# profiles/r03_k1_counters_p{14,20}.json, which bench.py reads for roofline.traffic and the VALU-issue bound.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/${1:-r03_counters}; mkdir -p $OUT
for P in ${2:-14 20}; do
  for set in FETCH_SIZE WRITE_SIZE "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_ATOMIC_sum"; do
    tag=$(echo $set | cut -d' ' -f1)
    rm -rf $OUT/raw
    timeout 600 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/raw -o x -- python3 scripts/quick_bench.py 10 50e6 4 40 $P > $OUT/quick_p${P}_$tag.txt 2>&1
    cp "$(find $OUT/raw -name '*counter_collection.csv' | head -1)" $OUT/counters_p${P}_$tag.csv
    rm -rf $OUT/raw
  done
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/raw -o st -- python3 scripts/quick_bench.py 10 50e6 4 40 $P > $OUT/quick_p${P}_stats.txt 2>&1
  cp "$(find $OUT/raw -name '*kernel_stats.csv' | head -1)" $OUT/kernel_stats_p$P.csv
  rm -rf $OUT/raw
  python3 scripts/make_counters_json.py $OUT $P $OUT/r03_k1_counters_p$P.json
done
