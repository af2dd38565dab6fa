# round-2 evidence: bench lines at log2m 14 / 18 / 20, kernel stats for each
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/${1:-r02_v1}
mkdir -p $OUT
for P in 14 18 20; do
  timeout 600 python bench.py --steps 10 --warmup 2 --log2m $P $( [ $P != 14 ] && echo --no-cpu-baseline ) > $OUT/bench_p$P.json 2> $OUT/bench_p$P.err
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats$P -o st -- python3 bench.py --steps 5 --warmup 1 --log2m $P --no-cpu-baseline --no-accuracy --no-secondary --no-ingest > /dev/null 2> $OUT/prof_p$P.err
  cp $(find $OUT/stats$P -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats_p$P.csv
  rm -rf $OUT/stats$P
done
for P in 16 17 19; do
  timeout 600 python bench.py --steps 10 --warmup 2 --log2m $P --no-cpu-baseline --no-accuracy --no-secondary --no-ingest > $OUT/bench_p$P.json 2> $OUT/bench_p$P.err
done
python3 - <<PY
import json
for p in (14, 16, 17, 18, 19, 20):
    d = json.load(open("$OUT/bench_p%d.json" % p))
    print(p, round(d["value"], 2), "Gbp/s", round(d["ms_per_step"], 2), "ms/step")
PY
