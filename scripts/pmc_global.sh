# L2 / atomic counters of the log2m >= 18 (registers in HBM) K1 path
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/pg
for set in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_ATOMIC_sum" "TCC_EA0_ATOMIC_sum TCP_TCC_ATOMIC_WITH_RET_REQ_sum SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY"; do
  tag=$(echo $set | cut -d' ' -f1)
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d gpurun_out/pg/$tag -o x -- python3 scripts/quick_bench.py 4 50e6 4 40 ${1:-18} > /dev/null 2>&1
  python3 scripts/pmc_summary.py $(find gpurun_out/pg/$tag -name "*counter_collection.csv" | head -1) 200e6 | grep -A6 "^sweep"
  rm -rf gpurun_out/pg/$tag
done
