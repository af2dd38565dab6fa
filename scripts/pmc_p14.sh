#!/bin/bash
# instruction counters of the headline (log2m 14) step per K1 class: the VALU-issue bound in bench.py uses them
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/${1:-r02_pmc_p14}; mkdir -p $OUT
timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $OUT/pmc -o pmc -- python3 scripts/quick_bench.py 10 50e6 > /dev/null 2>&1
python3 scripts/pmc_summary.py $(find $OUT/pmc -name "*counter_collection.csv" | head -1) > $OUT/pmc.txt
rm -rf $OUT/pmc
cat $OUT/pmc.txt
