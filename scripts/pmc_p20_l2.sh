cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r02_p20_l2; mkdir -p $OUT
timeout 400 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum --kernel-trace --output-format csv -d $OUT/raw1 -o x -- python3 scripts/quick_bench.py 10 50e6 4 40 20 > /dev/null 2>&1
python3 scripts/pmc_any.py $(find $OUT/raw1 -name "*counter_collection.csv" | head -1) > $OUT/l2.txt
rm -rf $OUT/raw1
grep -A5 "scatter\|replay\|sort_chunks" $OUT/l2.txt
