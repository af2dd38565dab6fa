#!/bin/bash
{
echo "== 10 x 50"; timeout 300 python scripts/files_probe.py 10 50e6
echo "== 64 x 5"; timeout 300 python scripts/files_probe.py 64 5e6
echo "== 10 x 50 p20 (512)"; P=20 timeout 300 python scripts/files_probe.py 10 50e6
echo "== 10 x 50 p20 256"; DD_BATCH_MB=256 P=20 timeout 300 python scripts/files_probe.py 10 50e6
echo "== 10 x 50 p20 128"; DD_BATCH_MB=128 P=20 timeout 300 python scripts/files_probe.py 10 50e6
echo "== trace p20"; P=20 DD_TRACE_FILES=1 timeout 300 python scripts/files_probe.py 10 50e6 2>&1 | tail -9
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/exp_ingest.txt
