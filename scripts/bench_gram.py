"""K2 all-pairs on the matrix cores, by unit shape (round 5): 64-row diagonal units (round 4, DD_GRAM_DIAG2=0) against 128-row ones
(n > 64), GRAM_ZEROS=0.01 [GRAM_MIN=v] sets every 100th register to 0 [v]: one more threshold per step of v.  Same slab, every variant's estimates compared.
  python scripts/bench_gram.py [LOG2M] [N] [K]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dandd_amd.engine import Engine, KERNEL_UNION

p = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n = int(sys.argv[2]) if len(sys.argv) > 2 else 128
K = int(sys.argv[3]) if len(sys.argv) > 3 else 31
m = 1 << p
eng = Engine(0, p, True)
rng = np.random.default_rng(0)
leaf = torch.empty((n, K, m), dtype=torch.uint8, device="cuda")
for i in range(n):   # registers distributed like a sketch of ~300 items per register
    u = rng.random((K, m), dtype=np.float32) + 1e-9
    leaf[i] = torch.from_numpy(np.clip(np.floor(np.log2(300.0) - np.log2(-np.log(u))) + 1, 0, 64 - p + 1).astype(np.uint8)).cuda()
if os.environ.get("GRAM_ZEROS"):   # small genomes leave registers empty: value 0 in every column, ~35 thresholds instead of ~29
    leaf[:, :, :: int(1 / float(os.environ["GRAM_ZEROS"]))] = int(os.environ.get("GRAM_MIN", "0"))
lo = leaf.amin(dim=(0, 2)).cpu().numpy().astype(int)
hi = leaf.amax(dim=(0, 2)).cpu().numpy().astype(int)
T = int((hi - lo).sum())
ns = (n + 63) // 64
blocks = ns * 3 + ns * (ns - 1) // 2 * 4
mfma = T * (m // 32) * blocks
first = None
times = {}
variants = [(d2, 0) for d2 in (0, 1) if not (d2 and n <= 64)]
for rnd in range(4):            # variants interleaved, four rounds in one process: the chip's clock moves with what ran before
    for d2, xcd in variants:
        os.environ["DD_GRAM_DIAG2"] = str(d2)
        out = eng.pairwise_device(leaf.data_ptr(), n, K)
        if first is None:
            first = out
        assert np.array_equal(out, first), (d2, xcd)
        eng.timing_enable(True)
        eng.timing_reset()
        for _ in range(5):
            eng.pairwise_device(leaf.data_ptr(), n, K)
        eng.synchronize()
        ms, _ = eng.timing_read(KERNEL_UNION)
        eng.timing_enable(False)
        times.setdefault((d2, xcd), []).append(ms / 5)
for (d2, xcd), ts in times.items():
    ms = float(np.median(ts))
    tops = 2.0 * 32 * 32 * 32 * mfma / (ms / 1e3) / 1e12
    print(f"log2m {p} n {n} K {K} thresholds {T}: diag2 {d2}: median {ms:8.3f} ms (rounds {' '.join(f'{t:.3f}' for t in ts)})  {tops:7.1f} TOP/s = {tops / 5000:.3f} of int8 dense", flush=True)
os.environ.pop("DD_GRAM_DIAG2")
if n * K * m <= (1 << 30):
    os.environ["DD_PAIRWISE_STREAM"] = "1"
    stream = eng.pairwise_device(leaf.data_ptr(), n, K)
    os.environ.pop("DD_PAIRWISE_STREAM")
    print("== streaming kernel:", bool(np.array_equal(stream, first)))
    assert np.array_equal(stream, first)
