#!/bin/bash
# instruction counters of the device inflate on 1 / 8 / 40 BGZF blocks: pmc_one_block.sh OUTNAME
OUT=gpurun_out/${1:-pmc_one}; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p $OUT
for pass in "SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVES" "SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
  tag=$(echo $pass | cut -d' ' -f1)
  timeout 600 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $OUT/$tag -o st -- python3 scripts/one_block_probe.py > $OUT/probe_$tag.txt 2>&1
  f=$(find $OUT/$tag -name "*counter_collection.csv" | head -1)
  python3 - <<PY
import csv, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open("$f")):
    if "inflate" in r["Kernel_Name"]:
        acc[(r["Dispatch_Id"], r["Grid_Size"])][r["Counter_Name"]] += float(r["Counter_Value"])
for k in sorted(acc, key=lambda t: int(t[0])):
    print("dispatch", k[0], "grid", k[1], dict(acc[k]))
PY
  rm -rf $OUT/$tag
done
