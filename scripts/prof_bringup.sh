# which HIP API calls a fresh process pays for before and inside its first dd_sketch_files call (log2m $1)
P=${1:-20}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export DANDD_NO_TORCH=1
OUT=gpurun_out/bringup_p$P; mkdir -p $OUT
[ -d /tmp/e2e/genomes ] || python3 scripts/e2e_cli.py 10 50 --dir /tmp/e2e --keep > /dev/null 2>&1
timeout 300 rocprofv3 --hip-trace --stats --output-format csv -d $OUT/t -o st -- python3 scripts/init_probe.py $P /tmp/e2e/genomes > $OUT/probe.txt 2>&1
grep -v amdgpu.ids $OUT/probe.txt | grep -E "ms" 
f=$(find $OUT/t -name "*hip_api_stats.csv" | head -1)
head -16 "$f" | cut -d, -f1-6
