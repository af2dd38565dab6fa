cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r02_p20_pmc; mkdir -p $OUT
timeout 400 rocprofv3 --pmc TCC_ATOMIC_sum TCC_EA0_ATOMIC_sum TCC_REQ_sum --kernel-trace --output-format csv -d $OUT/raw1 -o x -- python3 scripts/quick_bench.py 10 50e6 4 40 20 > /dev/null 2>&1
python3 scripts/pmc_any.py $(find $OUT/raw1 -name "*counter_collection.csv" | head -1) > $OUT/tcc.txt
rm -rf $OUT/raw1
grep -A3 "scatter\|replay\|sort_chunks" $OUT/tcc.txt
