#!/bin/bash
{
for T in twin notwin; do
  [ $T = notwin ] && export DD_NO_TWIN=1
  echo "== 10 x 50 $T"; timeout 300 python scripts/files_probe.py 10 50e6 | tail -4
  echo "== 64 x 5 $T"; timeout 300 python scripts/files_probe.py 64 5e6 | tail -4
  echo "== 10 x 50 p20 $T"; P=20 timeout 300 python scripts/files_probe.py 10 50e6 | tail -4
done
unset DD_NO_TWIN
echo "== 10 x 50 twin batch 64"; DD_BATCH_MB=64 timeout 300 python scripts/files_probe.py 10 50e6 | tail -3
echo "== 10 x 50 p20 twin batch 128"; DD_BATCH_MB=128 P=20 timeout 300 python scripts/files_probe.py 10 50e6 | tail -3
timeout 300 python -m pytest tests -m gpu -x -q -k "files or ingest" 2>&1 | tail -2
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/exp_ingest.txt
