"""Randomized parity sweep of the log2m >= 17 path (scatter + sort + replay): random log2m 16..20, k ranges, canonical
flag, batches of 1..4 genomes of random sizes and content, and random schedule knobs (first epoch, longest epoch,
capacity -> overflow path, record budget, the exact-set class on small genomes):
GPU registers of the batched call vs the oracle, bit for bit.   python scripts/fuzz_buckets.py [N] [SEED]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from dandd_amd.engine import Engine
from oracle import dd_oracle as orc

n_cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
engines = {}
alph = [np.frombuffer(b"ACGT", np.uint8), np.frombuffer(b"ACGTacgtN", np.uint8), np.frombuffer(b"ACGTACGTACGTRYKMn-* 0>@\r", np.uint8)]
KNOBS = ["DD_BUCKET_E0", "DD_BUCKET_EMAX", "DD_BUCKET_CAP", "DD_BIGMAP_ANY_SIZE", "DD_BUCKET_GB"]
t0 = time.time()
for it in range(n_cfg):
    for k in KNOBS:
        os.environ.pop(k, None)
    p = int(rng.choice([16, 17, 18, 19, 20]))      # (16: registers in LDS, the neighbour of the smallest record-path size)
    if rng.integers(0, 2):
        os.environ["DD_BUCKET_E0"] = str(int(rng.choice([1, 2, 3, 8])))
    if rng.integers(0, 2):
        os.environ["DD_BUCKET_EMAX"] = str(int(rng.choice([1, 2, 5])))
    if rng.integers(0, 4) == 0:
        os.environ["DD_BUCKET_CAP"] = str(int(rng.choice([1, 2, 7, 40, 75, 150])))
    if rng.integers(0, 2):
        os.environ["DD_BIGMAP_ANY_SIZE"] = "1"
    if rng.integers(0, 6) == 0:
        os.environ["DD_BUCKET_GB"] = "1"
    canon = bool(rng.integers(0, 2))
    k1, k2 = sorted(int(x) for x in rng.integers(1, 65, size=2))
    if k2 - k1 > 6:
        k2 = k1 + 6
    if rng.integers(0, 4) == 0:  # the class boundaries around the big-bitmap ks (10, 11 at log2m >= 19)
        k1 = int(rng.integers(8, 12))
        k2 = k1 + int(rng.integers(0, 4))
    fas = []
    for g in range(int(rng.integers(1, 5))):
        parts = []
        for r in range(int(rng.integers(0, 4))):
            parts.append(b">rec %d\n" % r if rng.integers(0, 4) else b">\n")
            a = alph[int(rng.integers(0, 3))]
            total = int(rng.choice([0, 1, 50, 1000, 70000, 200000, 400000]) * rng.random()) + int(rng.integers(0, 3))
            width = int(rng.choice([1, 7, 60, 61, 64, 80, 1000, 10 ** 9]))
            seq = rng.choice(a, size=total).tobytes()
            parts.append(b"\n".join(seq[i:i + width] for i in range(0, len(seq), width)) + (b"\n" if rng.integers(0, 2) else b""))
        fas.append(np.frombuffer(b"".join(parts), dtype=np.uint8))
    eng = engines.setdefault((p, canon), Engine(0, p, canon))
    bufs = [torch.from_numpy(f.copy()).cuda() if f.size else torch.empty(16, dtype=torch.uint8, device="cuda") for f in fas]
    K = k2 - k1 + 1
    regs = torch.empty((len(fas), K, eng.m), dtype=torch.uint8, device="cuda")
    eng.sketch_device([b.data_ptr() for b in bufs], [f.size for f in fas], k1, k2, regs.data_ptr())
    eng.synchronize()
    got = regs.cpu().numpy()
    for g, fa in enumerate(fas):
        want = orc.sketch_sweep(fa, k1, k2, p, canon)
        if not np.array_equal(got[g], want):
            bad = np.argwhere(got[g] != want)
            env = {k: os.environ[k] for k in KNOBS if k in os.environ}
            print(f"MISMATCH cfg {it}: p={p} canon={canon} k={k1}..{k2} genome {g}/{len(fas)} bytes={fa.size} knobs={env}: {bad.shape[0]} registers, first {bad[0]}")
            sys.exit(1)
print(f"{n_cfg} random configurations of the log2m >= 16 scatter/sort/replay path bit-exact in {time.time() - t0:.1f} s")
