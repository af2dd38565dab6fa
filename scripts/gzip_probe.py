"""One large .gz genome through dd_sketch_files: the parallel decoder (dd_inflate.h) against the serial one
(DD_NO_PARALLEL_GZIP=1), plain gzip -1 and BGZF.  usage: gzip_probe.py [Mbp] [log2m]   (development aid; the file is built
with Python's zlib at level 1, ~60 MB/s: most of this script's run time)"""
import os
import sys
import tempfile
import time
import zlib

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dandd_amd.engine import Engine, synth_size

mbp = float(sys.argv[1]) if len(sys.argv) > 1 else 1000.0
p = int(sys.argv[2]) if len(sys.argv) > 2 else 14
nb = int(mbp * 1e6)
eng = Engine(0, p, True)
n = synth_size(nb, 24)
buf = torch.empty(n + 16, dtype=torch.uint8, device="cuda")
eng.synth_fasta_device(0xD4ADD, 0, nb, 24, buf.data_ptr())
eng.synchronize()
raw = buf[:n].cpu().numpy().tobytes()
d = tempfile.mkdtemp(prefix="dd_gz_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
try:
    t0 = time.time()
    co = zlib.compressobj(1, zlib.DEFLATED, 31)
    plain = os.path.join(d, "one.fa.gz")
    with open(plain, "wb") as f:
        for a in range(0, n, 1 << 26):
            f.write(co.compress(raw[a:a + (1 << 26)]))
        f.write(co.flush())
    bg = os.path.join(d, "one.bgzf.fa.gz")
    with open(bg, "wb") as f:   # BGZF: <= 64 KiB members with a 'BC' extra field that holds the block size
        for a in list(range(0, n, 65280)) + [n]:
            part = raw[a:a + 65280] if a < n else b""
            c = zlib.compressobj(1, zlib.DEFLATED, -15)
            body = c.compress(part) + c.flush()
            bsize = len(body) + 25
            f.write(b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + bsize.to_bytes(2, "little") + body +
                    zlib.crc32(part).to_bytes(4, "little") + len(part).to_bytes(4, "little"))
    print(f"{mbp:g} Mbp: {n/1e6:.0f} MB of FASTA -> {os.path.getsize(plain)/1e6:.0f} MB gzip -1, {os.path.getsize(bg)/1e6:.0f} MB BGZF "
          f"(built in {time.time()-t0:.0f} s)")
    want = eng.sketch_buffer(np.frombuffer(raw, dtype=np.uint8), 20, 22)
    for name, path in (("gzip -1, one member", plain), ("BGZF", bg)):
        for env in ({}, {"DD_NO_PARALLEL_GZIP": "1"}):
            os.environ.pop("DD_NO_PARALLEL_GZIP", None)
            os.environ.update(env)
            best = None
            for _ in range(3):
                t0 = time.perf_counter()
                got = eng.sketch_files([path], 20, 22)
                dt = time.perf_counter() - t0
                best = dt if best is None else min(best, dt)
            assert np.array_equal(got[0], want), name
            print(f"  {name:22s} {'serial  ' if env else 'parallel'}: {best*1e3:8.1f} ms = {nb/best/1e9:6.2f} Gbp/s ({n/best/1e9:.2f} GB/s of FASTA)")
    os.environ.pop("DD_NO_PARALLEL_GZIP", None)
finally:
    import shutil
    shutil.rmtree(d, ignore_errors=True)
