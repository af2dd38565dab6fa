#!/bin/bash
# Counter evidence for K1 (round 4's profile_r04.sh with the round as a variable: ROUND=r05 by default): one rocprofv3 pass per counter set (FETCH_SIZE and WRITE_SIZE never share a pass; --pmc
# runs carry --kernel-trace only, MI355X_MICROARCH.md) plus a --kernel-trace --stats pass, over scripts/quick_bench.py.
#   profile_k1_counters.sh OUTDIR TAG NG NB KMIN KMAX LOG2M [NREC]  ->  gpurun_out/OUTDIR/${ROUND}_k1_counters_TAG.json + kernel_stats_TAG.csv
# bench.py reads profiles/r0[45]_k1_counters_<TAG>.json (copied there by hand; the newest round's first) for roofline.traffic and the VALU-issue bound.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
ROUND=${ROUND:-r05}
OUT=gpurun_out/$1; TAG=$2; NG=$3; NB=$4; KMIN=$5; KMAX=$6; P=$7; NREC=${8:-5}
mkdir -p $OUT
for set in FETCH_SIZE WRITE_SIZE "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_ATOMIC_sum"; do
  tag=$(echo $set | cut -d' ' -f1)
  rm -rf $OUT/raw
  timeout 900 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/raw -o x -- python3 scripts/quick_bench.py $NG $NB $KMIN $KMAX $P $NREC > $OUT/quick_${TAG}_$tag.txt 2>&1
  cp "$(find $OUT/raw -name '*counter_collection.csv' | head -1)" $OUT/counters_${TAG}_$tag.csv
  rm -rf $OUT/raw
done
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/raw -o st -- python3 scripts/quick_bench.py $NG $NB $KMIN $KMAX $P $NREC > $OUT/quick_${TAG}_stats.txt 2>&1
cp "$(find $OUT/raw -name '*kernel_stats.csv' | head -1)" $OUT/kernel_stats_$TAG.csv
rm -rf $OUT/raw
python3 scripts/make_counters_json.py $OUT $P $OUT/${ROUND}_k1_counters_$TAG.json $NG $NB $KMIN $KMAX $TAG
grep "iter 2" $OUT/quick_${TAG}_stats.txt
rm -f $OUT/counters_${TAG}_*.csv
