#!/bin/bash
# Round-6 evidence set, one box: the default bench line + sidecar, the rocprofv3 --kernel-trace --stats summary of the benchmarked
# command at log2m 14 and 20, K1 counter files (FETCH_SIZE / WRITE_SIZE / SQ / TCC in separate --pmc passes) for the headline
# workload, the world-1 RCCL line through torch.distributed and through the C ABI, the CLI end to end one-shot and through the server.
# usage: profile_round6.sh OUTNAME     (results under gpurun_out/OUTNAME; copy into profiles/ as r06_<v>_*)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/${1:-r06_v2}; mkdir -p $OUT
timeout 900 python3 bench.py --detail $OUT/bench_default_detail.json > $OUT/bench_default_line.json 2> $OUT/bench_default.err
for P in 14 20; do
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats$P -o st -- python3 bench.py --steps 5 --warmup 1 --log2m $P --no-cpu-baseline --no-accuracy --no-secondary --no-ingest --detail $OUT/bench_prof_p${P}_detail.json > $OUT/bench_prof_p$P.json 2> $OUT/bench_prof_p$P.err
  cp "$(find $OUT/stats$P -name '*kernel_stats.csv' | head -1)" $OUT/kernel_stats_p$P.csv
  rm -rf $OUT/stats$P
done
timeout 600 python3 bench.py --gpus 1 --force-dist --steps 10 --warmup 2 --no-cpu-baseline --no-accuracy --no-secondary --no-ingest --detail $OUT/bench_rccl_world1_detail.json > $OUT/bench_rccl_world1.json 2> $OUT/bench_rccl_world1.err
timeout 600 python3 bench.py --gpus 1 --force-dist --abi-comm --steps 10 --warmup 2 --no-cpu-baseline --no-accuracy --no-secondary --no-ingest --detail $OUT/bench_rccl_abi_world1_detail.json > $OUT/bench_rccl_abi_world1.json 2> $OUT/bench_rccl_abi_world1.err
{ python3 scripts/e2e_cli.py 10 50 --registers 14; python3 scripts/e2e_cli.py 10 50 --registers 14 --server; python3 scripts/e2e_cli.py 10 50 --registers 20; python3 scripts/e2e_cli.py 10 50 --registers 20 --server;
  python3 scripts/e2e_cli.py 64 5 --mink 10 --maxk 40; python3 scripts/e2e_cli.py 64 5 --mink 10 --maxk 40 --server;
  python3 scripts/e2e_cli.py 64 5 --registers 20 --mink 10 --maxk 40; python3 scripts/e2e_cli.py 64 5 --registers 20 --mink 10 --maxk 40 --server; } 2>/dev/null | grep workload > $OUT/e2e_cli.txt
# K1 counter files of the round's build (FETCH_SIZE / WRITE_SIZE / SQ / TCC in separate --pmc passes + a --kernel-trace --stats pass each)
ROUND=r06 bash scripts/profile_k1_counters.sh $(basename $OUT)/ctr p14 10 50e6 4 40 14 > $OUT/counters_p14.log 2>&1
ROUND=r06 bash scripts/profile_k1_counters.sh $(basename $OUT)/ctr 64x5_p20 64 5e6 4 40 20 > $OUT/counters_64x5_p20.log 2>&1
ROUND=r06 bash scripts/profile_k1_counters.sh $(basename $OUT)/ctr p20 10 50e6 4 40 20 > $OUT/counters_p20.log 2>&1
python3 - <<PY
import json, glob, os
for f in sorted(glob.glob("$OUT/bench_*.json")):
    if f.endswith("_detail.json"): continue
    try:
        d = json.load(open(f))
        print(os.path.basename(f), len(open(f).read()), "bytes", round(d["value"], 2), d["unit"], round(d["ms_per_step"], 2), "ms/step", "k1", round(d["roofline"]["kernel_ms_per_step"], 2),
              "frac_of_mix", (d["roofline"].get("valu_bound") or {}).get("frac_of_mix"), "coll", (d.get("collectives") or {}).get("backend"))
    except Exception as e:
        print(os.path.basename(f), "ERR", e)
PY
cut -c1-400 $OUT/e2e_cli.txt
