"""Randomized equality sweep of the K2 schedules (wider than the test suite): the Gram all-pairs kernel and the bit-plane
progressive scan against the streaming kernels, on random register slabs -- random log2m, n, K, orderings, value ranges
(narrow, wide, constant columns, empty sketches).   python scripts/fuzz_k2.py [N] [SEED]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from dandd_amd.engine import Engine

n_cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
engines = {}
t0 = time.time()
for it in range(n_cfg):
    p = int(rng.choice([12, 14, 16, 17, 18, 18, 19, 19, 20]))      # (the bit-plane scan runs from log2m 18 on)
    m, q = 1 << p, 64 - p
    n = int(rng.choice([2, 3, 5, 17, 30, 31, 32, 33, 63, 64, 65, 100, 129, 200]))
    K = int(rng.choice([1, 2, 3, 4, 9, 31]))
    if n * K * m > (1 << 28):       # (256 MB of registers per configuration at most)
        K = max(1, (1 << 28) // (n * m))
    if n * K * m > (1 << 28):
        n = max(2, (1 << 28) // (K * m))
    lo = int(rng.integers(0, 20))
    hi = int(rng.integers(lo, q + 2))
    kind = int(rng.integers(0, 4))
    if kind == 0:
        slab = rng.integers(lo, hi + 1, size=(n, K, m), dtype=np.uint8)
    elif kind == 1:
        slab = np.minimum(rng.geometric(0.5, size=(n, K, m)) + lo, q + 1).astype(np.uint8)
    elif kind == 2:
        slab = np.full((n, K, m), lo, dtype=np.uint8)
        slab[:, :, rng.integers(0, m, size=7)] = hi
    else:
        slab = np.minimum(rng.geometric(0.3, size=(n, K, m)), q + 1).astype(np.uint8)
        slab[rng.random((n, K, m)) < 0.3] = 0
        slab[int(rng.integers(n))] = 0
    eng = engines.setdefault(p, Engine(0, p, True))
    dev = torch.from_numpy(slab).cuda()
    os.environ.pop("DD_PAIRWISE_STREAM", None)
    gram = eng.pairwise_device(dev.data_ptr(), n, K)
    os.environ["DD_PAIRWISE_STREAM"] = "1"
    stream = eng.pairwise_device(dev.data_ptr(), n, K)
    os.environ.pop("DD_PAIRWISE_STREAM", None)
    if not np.array_equal(gram, stream):
        bad = np.argwhere(gram != stream)
        print(f"PAIRWISE MISMATCH cfg {it}: p={p} n={n} K={K} kind={kind} lo={lo} hi={hi}: {len(bad)} entries, first {bad[0]}")
        sys.exit(1)
    if n <= 32:
        no = int(rng.integers(1, 14))
        ords = np.stack([rng.integers(0, n, size=n) if rng.integers(0, 3) == 0 else rng.permutation(n) for _ in range(no)]).astype(np.int32)
        scan = eng.progressive_device(dev.data_ptr(), n, K, ords)      # (the bit-plane scan from log2m 18 on, the streaming kernel below)
        os.environ["DD_PROGRESSIVE_STREAM"] = "1"
        strm = eng.progressive_device(dev.data_ptr(), n, K, ords)
        os.environ.pop("DD_PROGRESSIVE_STREAM", None)
        if not np.array_equal(scan, strm):
            bad = np.argwhere(scan != strm)
            print(f"PROGRESSIVE MISMATCH cfg {it}: p={p} n={n} K={K} no={no} kind={kind} lo={lo} hi={hi}: {len(bad)} entries, first {bad[0]}")
            sys.exit(1)
print(f"{n_cfg} random K2 configurations: Gram == streaming all pairs, bit-plane scan == streaming progressive, in {time.time() - t0:.1f} s")
