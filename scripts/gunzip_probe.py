"""Single-member .gz files through dd_sketch_files with the device decoder (find block starts, decode pieces without their
history, resolve): registers against the plain bytes', strict (no host fallback), then timing device / host.
    python scripts/gunzip_probe.py [N] [MBP] [LEVEL] [LOG2M]"""
import os, sys, time, tempfile, zlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dandd_amd.engine import Engine, EngineError
from oracle import dd_oracle as orc
ng = int(sys.argv[1]) if len(sys.argv) > 1 else 2
nb = int(float(sys.argv[2]) * 1e6) if len(sys.argv) > 2 else 5_000_000
level = int(sys.argv[3]) if len(sys.argv) > 3 else 6
p = int(sys.argv[4]) if len(sys.argv) > 4 else 14
realistic = len(sys.argv) > 5 and sys.argv[5] == "realistic"   # GC 35 %, repeats, 2 % N in long runs, short contigs (dd_synth.hip's second generator)
d = tempfile.mkdtemp(dir="/dev/shm")
eng = Engine(0, p, True)
paths, raws = [], []
for g in range(ng):
    raw = (orc.synth_realistic(0xD4ADD, g, nb) if realistic else orc.synth_fasta(0xD4ADD, g, nb, 5)).tobytes()
    co = zlib.compressobj(level, zlib.DEFLATED, 31)
    data = co.compress(raw) + co.flush()
    q = os.path.join(d, f"g{g}.fa.gz")
    open(q, "wb").write(data)
    paths.append(q); raws.append(raw)
print(f"{ng} files of {nb / 1e6:.0f} Mbp, gzip -{level}: {len(data) / 1e6:.1f} MB compressed each")
os.environ["DD_INFLATE_STRICT"] = "1"
os.environ.setdefault("DD_GUNZIP_MIN_KB", "64")
try:
    got = eng.sketch_files(paths, 19, 21)
    for g in range(ng):
        want = eng.sketch_buffer(np.frombuffer(raws[g], np.uint8), 19, 21)
        print("file", g, "registers equal:", bool(np.array_equal(got[g], want)))
except EngineError as e:
    print("REFUSED:", str(e)[-120:])
os.environ.pop("DD_INFLATE_STRICT")
for mode in ("device", "host"):
    if mode == "host": os.environ["DD_NO_GPU_GUNZIP"] = "1"
    ts = []
    for r in range(6):
        t0 = time.perf_counter(); eng.sketch_files(paths, 4, 40); ts.append(time.perf_counter() - t0)
    ms = sorted(ts[2:])[len(ts[2:]) // 2] * 1e3
    print(f"{mode}: median {ng * nb / ms / 1e6:.2f} Gbp/s ({ms:.1f} ms)")
import shutil; shutil.rmtree(d)
