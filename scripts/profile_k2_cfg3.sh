#!/bin/bash
# HBM-side bytes of the all-pairs launch of `bench.py --config cfg3` (64 x 5 Mbp, k 2-32) at log2m 14 and 20:
# FETCH_SIZE and WRITE_SIZE in separate rocprofv3 passes -> profiles/r03_k2_counters_cfg3_p<P>.json (roofline_k2.traffic)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/${1:-r03_k2cfg3}; mkdir -p $OUT
for P in 14 20; do
  for set in FETCH_SIZE WRITE_SIZE; do
    rm -rf $OUT/raw
    timeout 600 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/raw -o x -- python3 bench.py --config cfg3 --log2m $P --steps 2 --warmup 1 --no-cpu-baseline > $OUT/bench_p${P}_$set.json 2> $OUT/bench_p${P}_$set.err
    cp "$(find $OUT/raw -name '*counter_collection.csv' | head -1)" $OUT/counters_p${P}_$set.csv
    rm -rf $OUT/raw
  done
  python3 - <<PY
import collections, csv, json, re
P = $P
def load(tag):
    acc, disp = collections.defaultdict(float), collections.defaultdict(set)
    for r in csv.DictReader(open("$OUT/counters_p%d_%s.csv" % (P, tag))):
        k = re.sub(r"\(anonymous namespace\)::|dd::|void ", "", r["Kernel_Name"]).split("(")[0]
        k = re.match(r"[A-Za-z0-9_]+", k).group(0)
        if r["Counter_Name"] == tag:
            acc[k] += float(r["Counter_Value"]) * 1024.0
            disp[k].add(r["Dispatch_Id"])
    return acc, disp
f, fd = load("FETCH_SIZE")
w, wd = load("WRITE_SIZE")
launches = len(fd["gram_kernel"])
out = {"workload": {"genomes": 64, "K": 31, "log2m": P, "what": "bench.py --config cfg3: dd_pairwise_device over the 64 x 31 x 2^%d register slab" % P},
       "made_by": "scripts/profile_k2_cfg3.sh (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes)",
       "note": "per dd_pairwise_device call; fetch = 2 x FETCH_SIZE (gfx950 wide-read correction: LDS-DMA dwordx4 and 16-byte loads)",
       "launches_seen": launches, "kernels": {}}
tf = tw = 0.0
for k in ("gram_range_init_kernel", "gram_range_kernel", "gram_kernel", "gram_finish_kernel", "mle_kernel"):
    n = max(1, len(fd[k]))
    per = launches if k != "mle_kernel" else n
    fb, wb = 2 * f[k] / max(1, len(fd[k])) , w[k] / max(1, len(wd[k]))
    if k == "mle_kernel":
        continue
    out["kernels"][k] = {"fetch_bytes": fb, "write_bytes": wb}
    tf += fb; tw += wb
out["bytes_per_launch"] = {"fetch": tf, "write": tw, "total": tf + tw}
json.dump(out, open("$OUT/r03_k2_counters_cfg3_p%d.json" % P, "w"), indent=1)
print(P, out["bytes_per_launch"])
PY
done
