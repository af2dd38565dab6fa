"""dd_sketch_files over ten 50 Mbp .gz files (one member each, zlib level argv[1]) with DD_TRACE_FILES on the fifth call: when each batch was issued and what it waited for."""
import os, sys, time, tempfile, zlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dandd_amd.engine import Engine
from oracle import dd_oracle as orc
level = int(sys.argv[1]) if len(sys.argv) > 1 else 1
d = tempfile.mkdtemp(dir="/dev/shm")
eng = Engine(0, 14, True)
paths = []
for g in range(10):
    raw = orc.synth_fasta(0xD4ADD, g, 50_000_000, 5).tobytes()
    co = zlib.compressobj(level, zlib.DEFLATED, 31)
    q = os.path.join(d, f"g{g}.fa.gz"); open(q, "wb").write(co.compress(raw) + co.flush()); paths.append(q)
for r in range(5):
    if r == 4: os.environ["DD_TRACE_FILES"] = "1"
    t0 = time.perf_counter(); eng.sketch_files(paths, 4, 40); print("call", r, (time.perf_counter() - t0) * 1e3, "ms", flush=True)
import shutil; shutil.rmtree(d)
