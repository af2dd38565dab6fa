#!/bin/bash
# Round 5: finder range size of the device gunzip (DD_GUNZIP_GUESS_KB) re-tuned now that the windows are composed in two levels
# (round 4 picked 32 / 64 / 128 KiB while one workgroup walked a file's pieces at 7 us each).  Writes gpurun_out/gunzip_ranges.txt
mkdir -p gpurun_out
OUT=gpurun_out/gunzip_ranges.txt
: > $OUT
for kb in 8 16 32 64; do
  for cfg in "10 50 1" "10 50 6" "1 400 1" "1 400 6" "64 5 6"; do
    echo "== DD_GUNZIP_GUESS_KB=$kb  gunzip_probe $cfg" | tee -a $OUT
    DD_GUNZIP_GUESS_KB=$kb timeout 600 python scripts/gunzip_probe.py $cfg 2>&1 | grep -E "device|host|REFUSED|False|compressed" | tee -a $OUT
  done
done
