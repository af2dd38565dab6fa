#!/bin/bash
# kernel stats + per-launch timeline of dd_sketch_files over single-member .gz files (device path): prof_gunzip.sh OUTNAME [N MBP LEVEL LOG2M]
OUT=gpurun_out/${1:-prof_gunzip}; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p $OUT
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o st -- python3 scripts/gunzip_probe.py "$@" > $OUT/probe.txt 2>&1
cp $(find $OUT/stats -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv; cp $(find $OUT/stats -name "*kernel_trace.csv" | head -1) $OUT/kernel_trace.csv
rm -rf $OUT/stats
grep -E "median|REFUSED|equal: False" $OUT/probe.txt
python3 - <<PY
import csv
rows = sorted(csv.DictReader(open("$OUT/kernel_trace.csv")), key=lambda r: int(r["Start_Timestamp"]))
names = ("find_starts", "inflate_kernel", "piece_offsets", "windows_kernel", "translate_kernel", "chunk_crc")
mine = [i for i, r in enumerate(rows) if any(n in r["Kernel_Name"] for n in names)]
# the last device call: walk back from the last of our kernels while gaps stay small
start = mine[-1]
for a, b in zip(reversed(mine[:-1]), reversed(mine[1:])):
    if int(rows[b]["Start_Timestamp"]) - int(rows[a]["End_Timestamp"]) > 40e6: break
    start = a
t0 = int(rows[start]["Start_Timestamp"])
for r in rows[start:]:
    n = r["Kernel_Name"]
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - t0) / 1e6
    if e - s > 0.5 or any(k in n for k in names):
        print(f"{s:8.2f} {e:8.2f} {e - s:7.2f}  q{r.get('Queue_Id', '?')} {n[:70]} grid {r.get('Grid_Size_X', r.get('Grid_Size'))}")
PY
