"""One-off scale check (not in the test suite: ~3 GB of HBM, ~1 s of GPU): a single 1 Gbp synthetic genome,
k 4..40, log2m 14.  No oracle at this size, so size-independent properties: sketch(whole) ==
max(sketch(first 2 records), sketch(last 3 records)) computed from separate buffers, determinism, and
monotone sanity of the cardinalities.  Exercises 64-bit token offsets (61 k chunks, 15 k tiles)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dandd_amd.engine import Engine, synth_size

nb = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000_000
nrec, kmin, kmax = 5, 4, 40
p = int(sys.argv[2]) if len(sys.argv) > 2 else 14
K, m = kmax - kmin + 1, 1 << p
eng = Engine(0, p, True)
n = synth_size(nb, nrec)
buf = torch.empty(n + 16, dtype=torch.uint8, device="cuda")
eng.synth_fasta_device(0xD4ADD, 0, nb, nrec, buf.data_ptr())
eng.synchronize()
per = nb // nrec
recsz = 16 + per + (per + 79) // 80
cut = 2 * recsz
assert int(buf[cut].item()) == ord(">") and int(buf[cut - 1].item()) == ord("\n")
part_a = buf[:cut].clone()
part_b = buf[cut:n].clone()
regs = torch.empty((4, K, m), dtype=torch.uint8, device="cuda")
t0 = time.time()
eng.sketch_device([buf.data_ptr()], [n], kmin, kmax, regs[0].data_ptr())
eng.synchronize()
dt = time.time() - t0
t1 = time.time()
eng.sketch_device([buf.data_ptr()], [n], kmin, kmax, regs[1].data_ptr())
eng.synchronize()
dt2 = time.time() - t1
eng.sketch_device([part_a.data_ptr(), part_b.data_ptr()], [cut, n - cut], kmin, kmax, regs[2].data_ptr())
eng.synchronize()
whole, again = regs[0].cpu().numpy(), regs[1].cpu().numpy()
parts = np.maximum(regs[2].cpu().numpy(), regs[3].cpu().numpy())
card = eng.card_batch_device(regs[0].data_ptr(), K)
print(f"log2m {p}: {nb/1e9:.2f} Gbp in {dt*1e3:.1f} ms = {nb/dt/1e9:.2f} Gbp/s (first call, includes workspace allocation); second call {dt2*1e3:.1f} ms = {nb/dt2/1e9:.2f} Gbp/s")
print("deterministic:", np.array_equal(whole, again))
print("union of parts == whole:", np.array_equal(parts, whole))
ks = np.arange(kmin, kmax + 1)
print("delta", (card / ks).max(), "argmax-k", ks[(card / ks).argmax()], "card[k=31]/nb", card[31 - kmin] / nb)
assert np.array_equal(whole, again) and np.array_equal(parts, whole)
assert 0.9 < card[31 - kmin] / nb < 1.1
print("OK")
