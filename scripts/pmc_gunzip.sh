#!/bin/bash
# SQ counters of the device gunzip kernels over scripts/gunzip_probe.py: pmc_gunzip.sh OUTNAME N MBP LEVEL
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/${1:-gz_pmc}; shift; mkdir -p $OUT; : > $OUT/pmc.txt
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/raw$i -o x -- python3 scripts/gunzip_probe.py "$@" > /dev/null 2>&1
  python3 scripts/pmc_summary.py $(find $OUT/raw$i -name "*counter_collection.csv" | head -1) | grep -A9 "^inflate_kernel\|^find_starts" >> $OUT/pmc.txt
  rm -rf $OUT/raw$i
done
cat $OUT/pmc.txt
