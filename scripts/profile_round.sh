set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/v8
timeout 600 python bench.py --steps 10 --warmup 2 > gpurun_out/v8/bench.json 2> gpurun_out/v8/bench.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/v8/stats -o st -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-accuracy > gpurun_out/v8/bench_prof.json 2> gpurun_out/v8/bench_prof.err
cp $(find gpurun_out/v8/stats -name "*kernel_stats.csv" | head -1) gpurun_out/v8/kernel_stats.csv
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/v8/fetch -o f -- python3 scripts/quick_bench.py 10 50e6 > /dev/null 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/v8/write -o w -- python3 scripts/quick_bench.py 10 50e6 > /dev/null 2>&1
python3 scripts/make_traffic.py gpurun_out/v8/fetch gpurun_out/v8/write gpurun_out/v8/traffic.json
timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD --kernel-trace --output-format csv -d gpurun_out/v8/pmc -o pmc -- python3 scripts/quick_bench.py 10 50e6 > /dev/null 2>&1
python3 scripts/pmc_summary.py $(find gpurun_out/v8/pmc -name "*counter_collection.csv" | head -1) > gpurun_out/v8/pmc.txt
rm -rf gpurun_out/v8/stats gpurun_out/v8/fetch gpurun_out/v8/write gpurun_out/v8/pmc
cat gpurun_out/v8/bench.json
head -12 gpurun_out/v8/kernel_stats.csv
