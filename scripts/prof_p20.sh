# per-kernel time of one cfg 2 shaped call at log2m $1 (default 20), then a PMC pass
set -x
P=${1:-20}
OUT=gpurun_out/${2:-prof_p$P}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p $OUT
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o st -- python3 scripts/quick_bench.py 10 50e6 4 40 $P > $OUT/quick.txt 2>&1
cp $(find $OUT/stats -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
cp $(find $OUT/stats -name "*kernel_trace.csv" | head -1) $OUT/kernel_trace.csv
rm -rf $OUT/stats
cat $OUT/quick.txt
head -14 $OUT/kernel_stats.csv
python3 - <<PY
import csv, collections
rows = list(csv.DictReader(open("$OUT/kernel_trace.csv")))
# last third of the trace = the last of the three timed iterations
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"][:60] for r in rows]
last = len(rows) - 1 - names[::-1].index(next(n for n in names if "pack_stats" in n))
for r in rows[last:last + 60]:
    print(f'{(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6:8.3f} ms  grid {r["Grid_Size_X"] if "Grid_Size_X" in r else r.get("Grid_Size")}  {r["Kernel_Name"][:70]}')
PY
