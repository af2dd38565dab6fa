#!/bin/bash
# kernel stats of one quick_bench call: prof_any.sh OUTNAME N MBP KMIN KMAX LOG2M
OUT=gpurun_out/$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p $OUT
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o st -- python3 scripts/quick_bench.py "$@" > $OUT/quick.txt 2>&1
cp $(find $OUT/stats -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
rm -rf $OUT/stats
grep iter $OUT/quick.txt
python3 - <<PY
import csv
for r in list(csv.DictReader(open("$OUT/kernel_stats.csv")))[:12]:
    print(f'{r["Name"][:80]:80s} calls {r["Calls"]:>4s} avg {float(r["AverageNs"])/1e6:8.3f} ms  total {float(r["TotalDurationNs"])/1e6:8.2f} ms')
PY
