import os, sys
sys.path.insert(0, os.getcwd())
import torch, bench
from dandd_amd.engine import Engine
eng = Engine(0, 14, True)
os.environ["DD_TRACE_FILES"] = "1"
import sys
ng, mbp = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (10, 50)
r = bench.ingest_probe(eng, ng, mbp * 1_000_000, 5, 4, 40, torch, gz="bgzf", reps=6)
print(r["value"], r["ms"])
