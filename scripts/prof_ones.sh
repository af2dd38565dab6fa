#!/bin/bash
# per-kernel times of the log2m >= 17 path with and without the rho = 1 bitmap (DD_FIRST_ONES): prof_ones.sh "64 5e6 4 40 20"
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/ones
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for o in 0 1; do
  export DD_FIRST_ONES=$o
  rm -rf gpurun_out/ones/prof
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ones/prof -o st -- python3 scripts/quick_bench.py ${1:-64 5e6 4 40 20} > /dev/null 2>&1
  cp "$(find gpurun_out/ones/prof -name '*kernel_stats.csv' | head -1)" gpurun_out/ones/kernel_stats_$o.csv
  rm -rf gpurun_out/ones/prof
  python3 - <<PY
import csv
print("DD_FIRST_ONES=$o")
for r in list(csv.DictReader(open("gpurun_out/ones/kernel_stats_$o.csv")))[:12]:
    print(f'{r["Name"][:110]:110s} calls {r["Calls"]:>4s} avg {float(r["AverageNs"])/1e6:8.3f} ms  total {float(r["TotalDurationNs"])/1e6:8.2f} ms')
PY
done
