"""Ad-hoc timing probe (not the contract bench): N synthetic genomes resident in HBM, one k-sweep."""
import sys, time
import numpy as np
import torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dandd_amd.engine import Engine, synth_size, KERNEL_PACK, KERNEL_SWEEP

ng = int(sys.argv[1]) if len(sys.argv) > 1 else 4
nb = int(float(sys.argv[2])) if len(sys.argv) > 2 else 50_000_000
kmin = int(sys.argv[3]) if len(sys.argv) > 3 else 4
kmax = int(sys.argv[4]) if len(sys.argv) > 4 else 40
p = int(sys.argv[5]) if len(sys.argv) > 5 else 14
nrec = int(sys.argv[6]) if len(sys.argv) > 6 else 5
eng = Engine(0, p, True)
K = kmax - kmin + 1
bufs = []
for g in range(ng):
    n = synth_size(nb, nrec)
    b = torch.empty(n + 16, dtype=torch.uint8, device="cuda")
    eng.synth_fasta_device(0xD4ADD, g, nb, nrec, b.data_ptr())
    bufs.append((b, n))
regs = torch.empty((ng, K, 1 << p), dtype=torch.uint8, device="cuda")
eng.synchronize()
eng.timing_enable(True)
for it in range(3):
    eng.timing_reset()
    t0 = time.time()
    eng.sketch_device([b.data_ptr() for b, _ in bufs], [n for _, n in bufs], kmin, kmax, regs.data_ptr())
    eng.synchronize()
    dt = time.time() - t0
    pk = eng.timing_read(KERNEL_PACK)
    sw = eng.timing_read(KERNEL_SWEEP)
    print(f"iter {it}: wall {dt*1e3:.2f} ms  pack {pk[0]:.3f} ms/{pk[1]}  sweep {sw[0]:.3f} ms/{sw[1]}  "
          f"Gbp/s {ng*nb/dt/1e9:.2f}  Gupd/s {ng*nb*K/dt/1e9:.1f}")
est = eng.card_batch_device(regs.data_ptr(), ng * K).reshape(ng, K)
ks = np.arange(kmin, kmax + 1)
d = est / ks
print("genome0 argmax-k", ks[d[0].argmax()], "delta", d[0].max())
