"""Where does dd_sketch_files spend its time?  (development probe)"""
import os, sys, time, tempfile
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dandd_amd.engine import Engine, synth_size
ng = int(sys.argv[1]) if len(sys.argv) > 1 else 64
nb = int(float(sys.argv[2])) if len(sys.argv) > 2 else 5_000_000
kmin, kmax, p = 10, 40, 14
eng = Engine(0, p, True)
d = tempfile.mkdtemp(prefix="dd_probe_")
n = synth_size(nb, 5)
buf = torch.empty(n + 16, dtype=torch.uint8, device="cuda")
paths = []
for g in range(ng):
    eng.synth_fasta_device(0xD4ADD, g, nb, 5, buf.data_ptr()); eng.synchronize()
    pth = os.path.join(d, f"g{g:03d}.fasta"); buf[:n].cpu().numpy().tofile(pth); paths.append(pth)
for nt in (0, 0, 2, 1):
    t0 = time.time(); regs = eng.sketch_files(paths, kmin, kmax, nt); dt = time.time() - t0
    print(f"sketch_files nthreads={nt}: {dt*1e3:.1f} ms  ({ng*nb/dt/1e9:.2f} Gbp/s)")
t0 = time.time(); datas = [np.fromfile(pth, dtype=np.uint8) for pth in paths]; t_read = time.time() - t0
t0 = time.time()
for a in datas: eng.sketch_buffer(a, kmin, kmax)
t_buf = time.time() - t0
print(f"python read {t_read*1e3:.1f} ms; {ng} x sketch_buffer {t_buf*1e3:.1f} ms ({t_buf/ng*1e3:.2f} ms each)")
# device-resident batch
dev = [torch.from_numpy(np.concatenate([a, np.zeros(16, np.uint8)])).cuda() for a in datas]
out = torch.empty((ng, kmax - kmin + 1, 1 << p), dtype=torch.uint8, device="cuda")
eng.synchronize(); torch.cuda.synchronize()
for _ in range(2):
    t0 = time.time(); eng.sketch_device([x.data_ptr() for x in dev], [a.size for a in datas], kmin, kmax, out.data_ptr()); eng.synchronize(); dt = time.time() - t0
    print(f"one batched sketch_device of {ng} genomes: {dt*1e3:.2f} ms")
import shutil; shutil.rmtree(d)
