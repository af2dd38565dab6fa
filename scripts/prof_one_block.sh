#!/bin/bash
# kernel durations of the device inflate on 1 / 8 / 40 BGZF blocks alone on the chip: prof_one_block.sh OUTNAME
OUT=gpurun_out/${1:-prof_one}; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p $OUT
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/stats -o st -- python3 scripts/one_block_probe.py > $OUT/probe.txt 2>&1
cp $(find $OUT/stats -name "*kernel_trace.csv" | head -1) $OUT/kernel_trace.csv
rm -rf $OUT/stats
python3 - <<PY
import csv
rows = [r for r in csv.DictReader(open("$OUT/kernel_trace.csv")) if "inflate" in r["Kernel_Name"]]
for r in rows:
    print("inflate launch: grid", r.get("Grid_Size_X", r.get("Grid_Size")), "ms", (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
PY
