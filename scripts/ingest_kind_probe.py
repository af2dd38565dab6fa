"""One ingest probe of bench.py alone (for rocprofv3 --kernel-trace --stats over it): python3 scripts/ingest_kind_probe.py KIND [NFILES] [MBP] [REPS]
KIND: plain | gzip1 | gzip6 | bgzf | fastq | members   (bench.ingest_probe's modes: dd_sketch_files over files in /dev/shm, k 4-40, log2m 14)"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from dandd_amd.engine import Engine
kind = sys.argv[1] if len(sys.argv) > 1 else "fastq"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 10
mbp = float(sys.argv[3]) if len(sys.argv) > 3 else 50
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 8
gz = {"plain": False, "gzip1": True, "gzip6": "gzip6", "bgzf": "bgzf", "fastq": "fastq", "members": "members"}[kind]
eng = Engine(0, 14, True)
r = bench.ingest_probe(eng, n, int(mbp * 1e6), 5, 4, 40, torch, gz=gz, reps=reps)
print(json.dumps({k: v for k, v in r.items() if k != "what"}))
