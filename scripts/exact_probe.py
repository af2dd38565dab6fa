"""Throughput of the exact distinct-k-mer counter (the KMC stand-in) on one HBM-resident genome."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dandd_amd.engine import Engine, synth_size
nb = int(float(sys.argv[1])) if len(sys.argv) > 1 else 50_000_000
eng = Engine(0, 14, True)
n = synth_size(nb, 5)
buf = torch.empty(n + 16, dtype=torch.uint8, device="cuda")
eng.synth_fasta_device(0xD4ADD, 0, nb, 5, buf.data_ptr()); eng.synchronize()
for k in (12, 12, 21, 31, 32, 33, 48, 64):
    t0 = time.time(); d = eng.exact_count_device([buf.data_ptr()], [n], k); dt = time.time() - t0
    print(f"k={k:2d}: {d:>12d} distinct in {dt*1e3:7.2f} ms  ({nb/dt/1e9:.2f} Gbp/s)")
