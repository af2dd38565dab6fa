# L2 / fabric counters of the log2m 20 path (scatter + sort + replay): is any global atomic left?
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/${1:-r02_p20_pmc}; mkdir -p $OUT; : > $OUT/pmc.txt
i=0
for set in "TCC_ATOMIC_sum TCC_EA0_ATOMIC_sum TCC_REQ_sum TCC_EA0_WRREQ_sum TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU"; do
  i=$((i+1))
  timeout 400 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/raw$i -o x -- python3 scripts/quick_bench.py 10 50e6 4 40 20 > /dev/null 2>&1
  python3 scripts/pmc_any.py $(find $OUT/raw$i -name "*counter_collection.csv" | head -1) >> $OUT/pmc.txt
  rm -rf $OUT/raw$i
done
DD_NO_BUCKETS=1 timeout 400 rocprofv3 --pmc TCC_ATOMIC_sum TCC_EA0_ATOMIC_sum TCC_REQ_sum --kernel-trace --output-format csv -d $OUT/raw9 -o x -- python3 scripts/quick_bench.py 10 50e6 4 40 20 > /dev/null 2>&1
echo "---- round 1's compare-and-swap path (DD_NO_BUCKETS=1), same call" >> $OUT/pmc.txt
python3 scripts/pmc_any.py $(find $OUT/raw9 -name "*counter_collection.csv" | head -1) sweep_kernel >> $OUT/pmc.txt
rm -rf $OUT/raw9
grep -A8 "scatter_kernel<1\|replay_kernel\|sort_chunks\|sweep_kernel<1" $OUT/pmc.txt | head -80
