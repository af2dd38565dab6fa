# SQ counters of the log2m >= 18 scatter/replay kernels (one cfg 2 shaped call, three iterations)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
P=${1:-20}
OUT=gpurun_out/${2:-pmc_scatter}
mkdir -p $OUT
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU" \
           "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_WAVES" \
           "TCC_REQ_sum TCC_ATOMIC_sum TCC_EA0_WRREQ_sum TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/raw$i -o x -- python3 scripts/quick_bench.py 10 50e6 4 40 $P > /dev/null 2>&1
  python3 scripts/pmc_any.py $(find $OUT/raw$i -name "*counter_collection.csv" | head -1) ${3:-scatter} >> $OUT/pmc.txt
  rm -rf $OUT/raw$i
done
cat $OUT/pmc.txt
