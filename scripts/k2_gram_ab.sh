#!/bin/bash
# K2 all-pairs A/B: Gram matrices on the matrix cores (default) against the streaming kernel (DD_PAIRWISE_STREAM=1),
# plus the per-kernel breakdown of the Gram path (rocprofv3 --kernel-trace --stats).  usage: k2_gram_ab.sh [ab|prof|both]
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/k2
what=${1:-both}
if [ "$what" != prof ]; then
for p in 14 20; do
  echo "== log2m $p, gram"; python3 scripts/bench_k2.py $p 2>/dev/null | grep -E "pairwise|progressive"
  echo "== log2m $p, stream"; DD_PAIRWISE_STREAM=1 python3 scripts/bench_k2.py $p 2>/dev/null | grep pairwise
done | tee gpurun_out/k2/ab.txt
fi
if [ "$what" != ab ]; then
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for p in 14 20; do
rm -rf gpurun_out/k2/prof$p
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/k2/prof$p -o st -- python3 scripts/bench_k2.py $p > /dev/null 2>&1
cp "$(find gpurun_out/k2/prof$p -name '*kernel_stats.csv' | head -1)" gpurun_out/k2/kernel_stats_p$p.csv
rm -rf gpurun_out/k2/prof$p
python3 - <<PY
import csv
print("log2m $p")
for r in list(csv.DictReader(open("gpurun_out/k2/kernel_stats_p$p.csv")))[:10]:
    print(f'{r["Name"][:90]:90s} calls {r["Calls"]:>4s} avg {float(r["AverageNs"])/1e6:8.3f} ms  total {float(r["TotalDurationNs"])/1e6:8.2f} ms')
PY
done
fi
