#!/bin/bash
# HBM-side bytes per kernel of a many-small-genomes call (64 x 5 Mbp, k 10..40, log2m $1): FETCH_SIZE and WRITE_SIZE in
# separate rocprofv3 passes (MI355X_MICROARCH.md: units of 64 B... as counted, FETCH x 2 on gfx950 for wide reads)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
P=${1:-20}; OUT=gpurun_out/pmc_small_p$P; mkdir -p $OUT; : > $OUT/traffic.txt
for set in FETCH_SIZE WRITE_SIZE; do
  timeout 400 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/raw_$set -o x -- python3 scripts/quick_bench.py 64 5e6 10 40 $P > /dev/null 2>&1
  python3 - "$(find $OUT/raw_$set -name '*counter_collection.csv' | head -1)" $set >> $OUT/traffic.txt <<'PY'
import csv, collections, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = collections.defaultdict(float); n = collections.defaultdict(int)
for r in rows:
    name = re.sub(r'\(anonymous namespace\)::|dd::|void ', '', r['Kernel_Name'])
    key = name.split('(')[0][:60]
    tot[key] += float(r['Counter_Value']); n[key] += 1
print(sys.argv[2], "summed over the 3 calls of the script (first call included), KiB-units as counted -> GB per call")
for k in sorted(tot, key=lambda k: -tot[k])[:8]:
    print(f"  {k:62s} launches {n[k]:4d}  {tot[k] * 1024 / 3 / 1e9:8.2f} GB per call (counter x 1 KiB)")
PY
  rm -rf $OUT/raw_$set
done
cat $OUT/traffic.txt
