"""dd_sketch_files over N BGZF files of MBP Mbp each: device inflate (dd_ginflate.hip) against the host decoder.
    python scripts/bgzf_probe.py [N] [MBP] [LOG2M]"""
import os, sys, time, tempfile, shutil
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from dandd_amd.engine import Engine
ng = int(sys.argv[1]) if len(sys.argv) > 1 else 10
nb = int(float(sys.argv[2]) * 1e6) if len(sys.argv) > 2 else 50_000_000
p = int(sys.argv[3]) if len(sys.argv) > 3 else 14
eng = Engine(0, p, True)
for mode in ("bgzf", False):
    for env in ((None, "1") if mode == "bgzf" else (None,)):
        if env: os.environ["DD_NO_GPU_INFLATE"] = env
        else: os.environ.pop("DD_NO_GPU_INFLATE", None)
        r = bench.ingest_probe(eng, ng, nb, 5, 4, 40, torch, gz=mode, reps=8)
        print(f"{'bgzf' if mode else 'plain'} {'host decoder' if env else ('device inflate' if mode else '')}: median {r['value']:.2f} Gbp/s ({r['ms']:.1f} ms), best {r['best_value']:.2f}")
