"""K2 timing probe (not the contract bench): the union schedules of BASELINE cfg 3 (kij: all pairs of
64 genomes) and cfg 4 (progressive: 10 orderings of 30 genomes) over HBM-resident register slabs.
Reports device time by HIP events and the algorithmic GB/s (register bytes read per union job)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dandd_amd.engine import Engine, KERNEL_UNION

p = int(sys.argv[1]) if len(sys.argv) > 1 else 14
only = sys.argv[2] if len(sys.argv) > 2 else ""
m = 1 << p
eng = Engine(0, p, True)
rng = np.random.default_rng(0)


def slab(n, K):
    # registers distributed like a sketch of ~300 items per register
    u = rng.random((n, K, m))
    r = np.clip(np.floor(np.log2(300.0) - np.log2(-np.log(u))) + 1, 0, 64 - p + 1).astype(np.uint8)
    return torch.from_numpy(r).cuda()


def timed(fn, reps=5):
    fn()
    eng.timing_enable(True)
    eng.timing_reset()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    eng.synchronize()
    wall = (time.perf_counter() - t0) / reps
    ms, n = eng.timing_read(KERNEL_UNION)
    eng.timing_enable(False)
    return wall * 1e3, ms / reps


# cfg 3: kij, 64 genomes, k 2..32
n, K = 64, 31
leaf = slab(n, K)
wall, dev = timed(lambda: eng.pairwise_device(leaf.data_ptr(), n, K))
pairs = n * (n + 1) // 2
bytes_read = (pairs * K) * 2 * m  # two inputs per pair job (the row operand stays in registers)
print(f"pairwise  n={n} K={K} p={p}: wall {wall:.2f} ms, union/hist kernels {dev:.3f} ms, "
      f"{pairs * K} union jobs, {bytes_read / dev / 1e6:.1f} GB/s algorithmic (2 inputs/job), "
      f"{pairs * K * m / dev / 1e6:.1f} GB/s counting the streamed operand only")

if only == "pairwise":
    sys.exit(0)
# cfg 4: progressive, 30 genomes, 10 orderings, k 4..40
n, K, no = 30, 37, 10
leaf = slab(n, K)
ords = np.stack([rng.permutation(n) for _ in range(no)]).astype(np.int32)
wall, dev = timed(lambda: eng.progressive_device(leaf.data_ptr(), n, K, ords))
jobs = no * n * K
print(f"progressive n={n} K={K} orderings={no} p={p}: wall {wall:.2f} ms, kernels {dev:.3f} ms, "
      f"{jobs} prefix unions, {jobs * m / dev / 1e6:.1f} GB/s (one new leaf array read per prefix)")

# N-way root union + cards of 100 genomes (cfg 5 shape per GPU)
n, K = 100, 37
leaf = slab(n, K)
out = torch.empty((K, m), dtype=torch.uint8, device="cuda")
ptrs = [leaf[i].data_ptr() for i in range(n)]
wall, dev = timed(lambda: eng.union_device(ptrs, K * m, out.data_ptr()))
print(f"union     n={n} K={K} p={p}: wall {wall:.3f} ms, kernel {dev:.3f} ms, {(n + 1) * K * m / dev / 1e6:.1f} GB/s")
wall, dev = timed(lambda: eng.card_batch_device(leaf.data_ptr(), n * K))
print(f"card      jobs={n * K} p={p}: wall {wall:.3f} ms, hist kernel {dev:.3f} ms, {n * K * m / dev / 1e6:.1f} GB/s")
