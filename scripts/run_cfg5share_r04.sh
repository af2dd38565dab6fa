mkdir -p gpurun_out/r04_cfg5
for P in 14 16 20; do
  timeout 1500 python bench.py --config cfg5share --log2m $P --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r04_cfg5/bench_cfg5share_p$P.json 2> gpurun_out/r04_cfg5/bench_cfg5share_p$P.err
  tail -c 600 gpurun_out/r04_cfg5/bench_cfg5share_p$P.json
  bash scripts/profile_r04.sh r04_cfg5 cfg5share_p$P 13 3e9 4 64 $P 24
done
