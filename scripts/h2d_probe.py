"""PCIe-inclusive rate of dd_sketch_buffer (host buffer -> registers on the host) vs the resident rate."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dandd_amd.engine import Engine, synth_size
nb = int(float(sys.argv[1])) if len(sys.argv) > 1 else 500_000_000
p = int(sys.argv[2]) if len(sys.argv) > 2 else 14
kmin, kmax = 4, 40
eng = Engine(0, p, True)
n = synth_size(nb, 5)
buf = torch.empty(n + 16, dtype=torch.uint8, device="cuda")
eng.synth_fasta_device(0xD4ADD, 0, nb, 5, buf.data_ptr()); eng.synchronize()
host = buf[:n].cpu().numpy().copy()
pinned = torch.empty(n, dtype=torch.uint8).pin_memory(); pinned.numpy()[:] = host
K = kmax - kmin + 1
regs = torch.empty((K, 1 << p), dtype=torch.uint8, device="cuda")
for name, arr in (("pageable", host), ("pageable", host), ("pinned", pinned.numpy()), ("pinned", pinned.numpy())):
    t0 = time.time(); eng.sketch_buffer(arr, kmin, kmax); dt = time.time() - t0
    print(f"sketch_buffer {name:8s}: {dt*1e3:7.1f} ms  {nb/dt/1e9:6.2f} Gbp/s  ({n/dt/1e9:.1f} GB/s of FASTA)")
for _ in range(2):
    t0 = time.time(); eng.sketch_device([buf.data_ptr()], [n], kmin, kmax, regs.data_ptr()); eng.synchronize(); dt = time.time() - t0
    print(f"sketch_device resident : {dt*1e3:7.1f} ms  {nb/dt/1e9:6.2f} Gbp/s")
t0 = time.time(); d = torch.from_numpy(host).cuda(); torch.cuda.synchronize(); dt = time.time() - t0
print(f"torch pageable H2D     : {dt*1e3:7.1f} ms  {n/dt/1e9:.1f} GB/s")
t0 = time.time(); d = pinned.cuda(non_blocking=True); torch.cuda.synchronize(); dt = time.time() - t0
print(f"torch pinned H2D       : {dt*1e3:7.1f} ms  {n/dt/1e9:.1f} GB/s")
