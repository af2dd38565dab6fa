#!/bin/bash
# Round-4 evidence set, one box, one go: bench lines of every BASELINE workload that fits one GPU, the rocprofv3
# --kernel-trace --stats summary of the benchmarked command at log2m 14 and 20, the world-1 RCCL line, the CLI end to end.
# usage: profile_round4.sh OUTNAME     (results under gpurun_out/OUTNAME; copy into profiles/ as r04_<v>_*)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/${1:-r04_v2}; mkdir -p $OUT
timeout 900 python3 bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
for P in 14 20; do
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats$P -o st -- python3 bench.py --steps 5 --warmup 1 --log2m $P --no-cpu-baseline --no-accuracy --no-secondary --no-ingest > $OUT/bench_prof_p$P.json 2> $OUT/bench_prof_p$P.err
  cp "$(find $OUT/stats$P -name '*kernel_stats.csv' | head -1)" $OUT/kernel_stats_p$P.csv
  rm -rf $OUT/stats$P
done
for P in 16 17 18 19 20; do
  timeout 600 python3 bench.py --steps 10 --warmup 2 --log2m $P --no-cpu-baseline --no-accuracy --no-secondary --no-ingest > $OUT/bench_p$P.json 2> $OUT/bench_p$P.err
done
timeout 600 python3 bench.py --config cfg3 --no-cpu-baseline > $OUT/bench_cfg3.json 2> $OUT/bench_cfg3.err
timeout 600 python3 bench.py --config cfg3 --log2m 20 --no-cpu-baseline > $OUT/bench_cfg3_p20.json 2> $OUT/bench_cfg3_p20.err
timeout 600 python3 bench.py --config cfg4share --no-cpu-baseline > $OUT/bench_cfg4share.json 2> $OUT/bench_cfg4share.err
timeout 600 python3 bench.py --config cfg4share --log2m 20 --no-cpu-baseline > $OUT/bench_cfg4share_p20.json 2> $OUT/bench_cfg4share_p20.err
for P in 14 16 20; do
  timeout 900 python3 bench.py --config cfg5share --log2m $P --steps 2 --warmup 1 --no-cpu-baseline > $OUT/bench_cfg5share_p$P.json 2> $OUT/bench_cfg5share_p$P.err
done
{ python3 scripts/bgzf_probe.py 10 50 14; python3 scripts/bgzf_probe.py 64 5 14; python3 scripts/bgzf_probe.py 3 300 14; } 2>/dev/null | grep -E "bgzf|plain" > $OUT/bgzf_probe.txt
{ for a in "10 50 6" "10 50 1" "64 5 6" "1 400 6" "3 300 6"; do echo "== $a"; python3 scripts/gunzip_probe.py $a 2>/dev/null | grep -E "files of|median|REFUSED|False"; done; } > $OUT/gunzip_probe.txt
timeout 600 python3 bench.py --gpus 1 --force-dist --steps 10 --warmup 2 --no-cpu-baseline --no-accuracy --no-secondary --no-ingest > $OUT/bench_rccl_world1.json 2> $OUT/bench_rccl_world1.err
{ python3 scripts/e2e_cli.py 10 50 --registers 14; python3 scripts/e2e_cli.py 10 50 --registers 20; python3 scripts/e2e_cli.py 64 5 --mink 10 --maxk 40; } 2>/dev/null | grep workload > $OUT/e2e_cli.txt
python3 - <<PY
import json, glob, os
for f in sorted(glob.glob("$OUT/bench_*.json")):
    try:
        d = json.load(open(f))
        k2 = d.get("roofline_k2") or {}
        print(os.path.basename(f), round(d["value"], 2), d["unit"], round(d["ms_per_step"], 2), "ms/step", "k1", round(d["roofline"]["kernel_ms_per_step"], 2),
              "k2", k2.get("ms"), "coll", d.get("collectives", {}).get("backend"))
    except Exception as e:
        print(os.path.basename(f), "ERR", e)
PY
cat $OUT/e2e_cli.txt | cut -c1-330
cat $OUT/bgzf_probe.txt $OUT/gunzip_probe.txt
