"""First dd_sketch_files call of a fresh context over N single-member .gz files: device path against host decoder
(the one-shot CLI pays this call, allocations included).  python scripts/gunzip_first_call.py [N] [MBP]"""
import os, sys, time, tempfile, zlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dandd_amd.engine import Engine
from oracle import dd_oracle as orc
ng = int(sys.argv[1]) if len(sys.argv) > 1 else 10
nb = int(float(sys.argv[2]) * 1e6) if len(sys.argv) > 2 else 50_000_000
d = tempfile.mkdtemp(dir="/dev/shm")
paths = []
for g in range(ng):
    raw = orc.synth_fasta(0xD4ADD, g, nb, 5).tobytes()
    co = zlib.compressobj(6, zlib.DEFLATED, 31)
    q = os.path.join(d, f"g{g}.fa.gz")
    open(q, "wb").write(co.compress(raw) + co.flush())
    paths.append(q)
warm = Engine(0, 14, True); warm.sketch_buffer(np.frombuffer(raw[:100000], np.uint8), 19, 21)   # HIP and the kernels' code objects are up
for rep in range(2):
    for mode in ("device", "host"):
        if mode == "host": os.environ["DD_NO_GPU_GUNZIP"] = "1"
        else: os.environ.pop("DD_NO_GPU_GUNZIP", None)
        eng = Engine(0, 14, True)
        t0 = time.perf_counter(); eng.sketch_files(paths, 4, 40); t1 = time.perf_counter(); eng.sketch_files(paths, 4, 40); t2 = time.perf_counter()
        print(f"{mode}: first call {1e3 * (t1 - t0):.1f} ms, second {1e3 * (t2 - t1):.1f} ms")
        eng.close()
import shutil; shutil.rmtree(d)
