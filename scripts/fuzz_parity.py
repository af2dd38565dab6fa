"""One-off randomized parity sweep (wider than the test suite): random log2m, k ranges, canonical flag,
genome sizes and byte content, GPU registers vs the oracle, bit for bit.   python scripts/fuzz_parity.py [N] [SEED]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa: F401
from dandd_amd.engine import Engine
from oracle import dd_oracle as orc

n_cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
engines = {}
alph = [np.frombuffer(b"ACGT", np.uint8), np.frombuffer(b"ACGTacgtN", np.uint8), np.frombuffer(b"ACGTACGTACGTRYKMn-* 0>", np.uint8),
        # kseq's record rules (oracle/POLICIES.md P10): '@' and '+' at line starts, '\r' inside and at the end of lines
        np.frombuffer(b"ACGTACGTACGTACGTacgtN@+\r>", np.uint8)]
t0 = time.time()
for it in range(n_cfg):
    p = int(rng.choice([4, 7, 10, 12, 13, 14, 15, 16, 17, 18, 19, 20]))
    canon = bool(rng.integers(0, 2))
    k1, k2 = sorted(int(x) for x in rng.integers(1, 65, size=2))
    if k2 - k1 > 24:
        k2 = k1 + 24
    parts = []
    nrec = int(rng.integers(0, 6))
    if rng.integers(0, 6) == 0:     # text in front of the first header
        parts.append(rng.choice(alph[1], size=int(rng.integers(1, 300))).tobytes() + (b"\n" if rng.integers(0, 2) else b""))
    for r in range(nrec):
        if rng.integers(0, 8) == 0:  # a FASTQ record (the host rewrites these before K0: dd_io.h)
            q = rng.choice(alph[1], size=int(rng.integers(0, 400))).tobytes()
            w = int(rng.choice([60, 10 ** 9]))
            body = b"\n".join(q[i:i + w] for i in range(0, len(q), w))
            qual = bytes(rng.choice(np.frombuffer(b"I@>+#5", np.uint8), size=len(q)))
            parts.append(b"@read %d\n" % r + body + b"\n+\n" + b"\n".join(qual[i:i + w] for i in range(0, len(qual), w)) + b"\n")
            continue
        parts.append((b">rec %d\n" % r if rng.integers(0, 4) else b">\n") if rng.integers(0, 8) else b"@rec %d\n" % r)
        a = alph[int(rng.integers(0, 4))]
        total = int(rng.choice([0, 1, 50, 1000, 70000, 300000]) * rng.random()) + int(rng.integers(0, 3))
        width = int(rng.choice([1, 7, 60, 61, 64, 80, 1000, 10 ** 9]))
        seq = rng.choice(a, size=total).tobytes()
        parts.append(b"\n".join(seq[i:i + width] for i in range(0, len(seq), width)) + (b"\n" if rng.integers(0, 2) else b""))
    fa = np.frombuffer(b"".join(parts), dtype=np.uint8)
    eng = engines.setdefault((p, canon), Engine(0, p, canon))
    got = eng.sketch_buffer(fa, k1, k2)
    want = orc.sketch_sweep(fa, k1, k2, p, canon)
    if not np.array_equal(got, want):
        bad = np.argwhere(got != want)
        print(f"MISMATCH cfg {it}: p={p} canon={canon} k={k1}..{k2} bytes={fa.size}: {bad.shape[0]} registers, first {bad[0]}")
        np.save(f"gpurun_out/fuzz_fail_{it}.npy", fa)
        sys.exit(1)
print(f"{n_cfg} random configurations bit-exact in {time.time() - t0:.1f} s")
