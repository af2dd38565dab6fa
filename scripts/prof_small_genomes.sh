# per-kernel time of a many-small-genomes call (64 x 5 Mbp, k 10..40) at log2m $1 (default 20)
set -x
P=${1:-20}
OUT=gpurun_out/${2:-prof_small_p$P}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p $OUT
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o st -- python3 scripts/quick_bench.py 64 5e6 10 40 $P > $OUT/quick.txt 2>&1
cp $(find $OUT/stats -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
cp $(find $OUT/stats -name "*kernel_trace.csv" | head -1) $OUT/kernel_trace.csv
rm -rf $OUT/stats
cat $OUT/quick.txt
python3 - <<PY
import csv
rows = list(csv.DictReader(open("$OUT/kernel_trace.csv")))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
last = len(rows) - 1 - names[::-1].index(next(n for n in names if "pack_stats" in n))
t0 = int(rows[last]["Start_Timestamp"])
for r in rows[last:last + 80]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f'{(s - t0) / 1e6:8.3f} +{(e - s) / 1e6:8.3f} ms  grid {r.get("Grid_Size_X", r.get("Grid_Size"))} wg {r.get("Workgroup_Size_X", r.get("Workgroup_Size"))} {r["Kernel_Name"][:90]}')
PY
