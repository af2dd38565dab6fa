#!/bin/bash
# kernel stats of one ingest probe: prof_ingest_kind.sh OUTNAME KIND [NFILES] [MBP] [REPS]
OUT=gpurun_out/$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p $OUT
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o st -- python3 scripts/ingest_kind_probe.py "$@" > $OUT/probe.txt 2>&1
cp $(find $OUT/stats -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
rm -rf $OUT/stats
tail -1 $OUT/probe.txt
python3 - <<PY
import csv
for r in list(csv.DictReader(open("$OUT/kernel_stats.csv")))[:16]:
    print(f'{r["Name"][:90]:90s} calls {r["Calls"]:>4s} avg {float(r["AverageNs"])/1e6:8.3f} ms  total {float(r["TotalDurationNs"])/1e6:8.2f} ms')
PY
