# HBM traffic counters of one bench step (separate --pmc passes, as MI355X_MICROARCH.md prescribes)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/v8
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/v8/fetch -o f -- python3 scripts/quick_bench.py 10 50e6 > /dev/null 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/v8/write -o w -- python3 scripts/quick_bench.py 10 50e6 > /dev/null 2>&1
python3 scripts/make_traffic.py gpurun_out/v8/fetch gpurun_out/v8/write gpurun_out/v8/traffic.json
rm -rf gpurun_out/v8/fetch gpurun_out/v8/write
cat gpurun_out/v8/traffic.json | head -50
