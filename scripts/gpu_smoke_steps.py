"""Staged GPU smoke: each step logs before/after so a hang or crash is attributable."""
import sys, time
import numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
T0 = time.time()
def log(*a):
    print(f"[{time.time()-T0:7.2f}s]", *a, flush=True)
log("import torch"); import torch
log("cuda available", torch.cuda.is_available(), torch.cuda.get_device_name(0))
from oracle import dd_oracle as orc
from dandd_amd.engine import Engine, synth_size
log("create engine p=10"); e10 = Engine(0, 10, True)
fa = orc.synth_fasta(0xD4ADD, 0, 1000, 1)
log("sketch 1kb k10..12"); got = e10.sketch_buffer(fa, 10, 12); want = orc.sketch_sweep(fa, 10, 12, 10)
log("  equal:", np.array_equal(got, want), "nonzero", int((got != 0).sum()), int((want != 0).sum()))
log("union"); u = e10.union([got, want]); log("  equal:", np.array_equal(u, want))
log("card"); c = e10.card(got[0]); log("  ", c, orc.card(want[0]))
fa = orc.synth_fasta(0xD4ADD, 1, 20000, 2)
log("sketch 20kb k1..64"); got = e10.sketch_buffer(fa, 1, 64); want = orc.sketch_sweep(fa, 1, 64, 10)
log("  equal:", np.array_equal(got, want), "mismatching rows:", np.where((got != want).any(axis=1))[0][:10])
leaf = np.stack([orc.sketch_sweep(orc.synth_fasta(0xD4ADD, g, 5000, 1), 8, 9, 10) for g in range(3)])
log("progressive"); pr = e10.progressive(leaf, np.array([[0, 1, 2], [2, 1, 0]], dtype=np.int32)); log("  ", pr[0, :, 0])
log("pairwise"); pw = e10.pairwise(leaf); log("  ", pw[:, :, 0])
log("create engine p=14"); e14 = Engine(0, 14, True)
fa = orc.synth_fasta(0xD4ADD, 0, 300000, 3)
log("sketch 300kb k4..40"); t = time.time(); got = e14.sketch_buffer(fa, 4, 40); dt = time.time() - t
want = orc.sketch_sweep(fa, 4, 40, 14)
log(f"  equal: {np.array_equal(got, want)}  ({dt*1e3:.1f} ms) mismatching rows:", np.where((got != want).any(axis=1))[0][:10])
log("create engine p=20"); e20 = Engine(0, 20, True)
fa = orc.synth_fasta(0xD4ADD, 2, 50000, 2)
log("sketch 50kb k14..18 p=20"); got = e20.sketch_buffer(fa, 14, 18); want = orc.sketch_sweep(fa, 14, 18, 20)
log("  equal:", np.array_equal(got, want))
log("device synth"); n = synth_size(12345, 4); b = torch.empty(n, dtype=torch.uint8, device="cuda")
e14.synth_fasta_device(0xD4ADD, 3, 12345, 4, b.data_ptr()); e14.synchronize()
log("  equal:", np.array_equal(b.cpu().numpy(), orc.synth_fasta(0xD4ADD, 3, 12345, 4)))
log("done")
