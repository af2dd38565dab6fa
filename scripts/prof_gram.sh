#!/bin/bash
# per-kernel times of the all-pairs path at n = 128 (log2m 20, K 31): 64-row against 128-row diagonal units
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/gram
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for d2 in 0 1; do
  export DD_GRAM_DIAG2=$d2
  rm -rf gpurun_out/gram/prof
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/gram/prof -o st -- python3 scripts/bench_gram_one.py ${1:-128} > /dev/null 2>&1
  cp "$(find gpurun_out/gram/prof -name '*kernel_stats.csv' | head -1)" gpurun_out/gram/kernel_stats_d2_$d2.csv
  rm -rf gpurun_out/gram/prof
  python3 - <<PY
import csv
print("DD_GRAM_DIAG2=$d2")
for r in list(csv.DictReader(open("gpurun_out/gram/kernel_stats_d2_$d2.csv")))[:9]:
    print(f'{r["Name"][:100]:100s} calls {r["Calls"]:>4s} avg {float(r["AverageNs"])/1e6:8.3f} ms  total {float(r["TotalDurationNs"])/1e6:8.2f} ms')
PY
done
