#!/bin/bash
# Round 5: the other BASELINE workloads that fit one GPU, one line each (compact line + sidecar), and fresh fuzz seeds.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/${1:-r05_cfg}; mkdir -p $OUT
for c in "cfg3 14" "cfg3 20" "cfg4share 14" "cfg4share 20"; do set -- $c
  timeout 600 python3 bench.py --config $1 --log2m $2 --no-cpu-baseline --detail $OUT/bench_$1_p$2_detail.json > $OUT/bench_$1_p$2.json 2> $OUT/bench_$1_p$2.err
done
timeout 900 python3 bench.py --config cfg5share --log2m 14 --steps 2 --warmup 1 --no-cpu-baseline --detail $OUT/bench_cfg5share_p14_detail.json > $OUT/bench_cfg5share_p14.json 2> $OUT/bench_cfg5share_p14.err
for P in 16 17 18 19; do
  timeout 600 python3 bench.py --steps 10 --warmup 2 --log2m $P --no-cpu-baseline --no-accuracy --no-secondary --no-ingest --detail $OUT/bench_p${P}_detail.json > $OUT/bench_p$P.json 2> $OUT/bench_p$P.err
done
python3 - <<PY
import json, glob, os
for f in sorted(glob.glob("$OUT/bench_*.json")):
    if f.endswith("_detail.json"): continue
    try:
        d = json.load(open(f)); k2 = d.get("roofline_k2") or {}
        print(os.path.basename(f), round(d["value"], 2), d["unit"], round(d["ms_per_step"], 2), "ms/step", "k1", round(d["roofline"]["kernel_ms_per_step"], 2), "k2", k2.get("path"), k2.get("ms"), k2.get("frac"), d.get("accuracy_subsample"))
    except Exception as e:
        print(os.path.basename(f), "ERR", e)
PY
