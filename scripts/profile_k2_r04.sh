#!/bin/bash
# HBM-side bytes of the K2 schedule launch of a bench.py config (round 4: cfg4share = dd_progressive_device over 8 x 31 sketches,
# 10 orderings; cfg3 = dd_pairwise_device) at log2m 14 and 20: FETCH_SIZE and WRITE_SIZE in separate rocprofv3 passes
#   profile_k2_r04.sh OUTDIR CONFIG  ->  gpurun_out/OUTDIR/r04_k2_counters_CONFIG_p<P>.json  (bench.py: roofline_k2.traffic)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/${1:-r04_k2}; CFG=${2:-cfg4share}; mkdir -p $OUT
for P in 14 20; do
  for set in FETCH_SIZE WRITE_SIZE; do
    rm -rf $OUT/raw
    timeout 900 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/raw -o x -- python3 bench.py --config $CFG --log2m $P --steps 2 --warmup 1 --no-cpu-baseline > $OUT/bench_${CFG}_p${P}_$set.json 2> $OUT/bench_${CFG}_p${P}_$set.err
    cp "$(find $OUT/raw -name '*counter_collection.csv' | head -1)" $OUT/counters_${CFG}_p${P}_$set.csv
    rm -rf $OUT/raw
  done
  python3 - <<PY
import collections, csv, json, re
P, CFG = $P, "$CFG"
bench = json.loads(open("$OUT/bench_%s_p%d_FETCH_SIZE.json" % (CFG, P)).read().strip().splitlines()[-1])
k2 = bench["roofline_k2"]
def load(tag):
    acc, disp = collections.defaultdict(float), collections.defaultdict(set)
    for r in csv.DictReader(open("$OUT/counters_%s_p%d_%s.csv" % (CFG, P, tag))):
        k = re.sub(r"\(anonymous namespace\)::|dd::|void ", "", r["Kernel_Name"]).split("(")[0]
        k = re.match(r"[A-Za-z0-9_]+", k).group(0)
        if r["Counter_Name"] == tag:
            acc[k] += float(r["Counter_Value"]) * 1024.0
            disp[k].add(r["Dispatch_Id"])
    return acc, disp
f, fd = load("FETCH_SIZE")
w, wd = load("WRITE_SIZE")
kernels = {"progressive_pscan": ("gram_range_init_kernel", "gram_range_kernel", "pscan_kernel", "pscan_finish_kernel"),
           "progressive_stream": ("progressive_kernel",),
           "pairwise_gram": ("gram_range_init_kernel", "gram_range_kernel", "gram_kernel", "gram_finish_kernel"),
           "pairwise_stream": ("pairwise_kernel",)}[k2["path"]]
cfgd = bench["config"]
n = cfgd["genomes_per_gpu"]
out = {"workload": {"genomes": n, "K": cfgd["kmax"] - cfgd["kmin"] + 1, "log2m": P, "path": k2["path"],
                    "what": "bench.py --config %s: the K2 schedule launch (%s) over the %d x %d x 2^%d register slab" % (CFG, k2["path"], n, cfgd["kmax"] - cfgd["kmin"] + 1, P)},
       "made_by": "scripts/profile_k2_r04.sh (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes)",
       "note": "per call of the schedule; fetch = 2 x FETCH_SIZE (gfx950 wide-read correction for 16-byte-per-lane reads)", "kernels": {}}
tf = tw = 0.0
for k in kernels:
    fb, wb = 2 * f[k] / max(1, len(fd[k])), w[k] / max(1, len(wd[k]))
    out["kernels"][k] = {"fetch_bytes": fb, "write_bytes": wb, "launches_seen": len(fd[k])}
    tf += fb; tw += wb
out["bytes_per_launch"] = {"fetch": tf, "write": tw, "total": tf + tw}
json.dump(out, open("$OUT/r04_k2_counters_%s_p%d.json" % (CFG, P), "w"), indent=1)
print(CFG, P, k2["path"], k2["ms"], out["bytes_per_launch"])
PY
  rm -f $OUT/counters_${CFG}_p${P}_*.csv
done
