#!/bin/bash
# kernel stats of dd_sketch_files over BGZF files (device inflate): prof_bgzf.sh OUTNAME [N MBP LOG2M]
OUT=gpurun_out/${1:-prof_bgzf}; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p $OUT
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o st -- python3 scripts/bgzf_probe.py "$@" > $OUT/probe.txt 2>&1
cp $(find $OUT/stats -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv; cp $(find $OUT/stats -name "*kernel_trace.csv" | head -1) $OUT/kernel_trace.csv
rm -rf $OUT/stats
tail -3 $OUT/probe.txt
python3 - <<PY
import csv
for r in list(csv.DictReader(open("$OUT/kernel_stats.csv")))[:8]:
    print(f'{r["Name"][:70]:70s} calls {r["Calls"]:>5s} avg {float(r["AverageNs"])/1e6:8.3f} ms  total {float(r["TotalDurationNs"])/1e6:9.2f} ms')
PY
python3 - <<PY
import csv, glob
f = glob.glob("$OUT/kernel_trace.csv")
PY
python3 - <<PY
import csv
rows = [r for r in csv.DictReader(open("$OUT/kernel_trace.csv")) if "inflate" in r["Kernel_Name"]]
for r in rows[-6:]:
    print("inflate launch: grid", r.get("Grid_Size_X", r.get("Grid_Size")), "wg", r.get("Workgroup_Size_X", r.get("Workgroup_Size")), "lds", r.get("LDS_Block_Size", r.get("LDS_Block_Size_v", "?")), "ms", (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
PY
