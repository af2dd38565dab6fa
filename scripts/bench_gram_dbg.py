"""diagnostic (results wrong by construction): the diagonal Gram kernel with one ingredient taken out at a time"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dandd_amd.engine import Engine, KERNEL_UNION
p, n, K = 20, int(sys.argv[1]) if len(sys.argv) > 1 else 64, 31
m = 1 << p
eng = Engine(0, p, True)
rng = np.random.default_rng(0)
leaf = torch.empty((n, K, m), dtype=torch.uint8, device="cuda")
for i in range(n):
    u = rng.random((K, m), dtype=np.float32) + 1e-9
    leaf[i] = torch.from_numpy(np.clip(np.floor(np.log2(300.0) - np.log2(-np.log(u))) + 1, 0, 64 - p + 1).astype(np.uint8)).cuda()
names = {0: "as shipped", 1: "no barrier", 2: "no LDS reads", 3: "no DMA, no vm waits", 4: "no thresholding"}
os.environ["DD_GRAM_DIAG2"] = sys.argv[2] if len(sys.argv) > 2 else "1"
for rep in range(2):
    for dbg in (0, 1, 2, 3, 4):
        os.environ["DD_GRAM_DBG"] = str(dbg)
        eng.pairwise_device(leaf.data_ptr(), n, K)
        eng.timing_enable(True); eng.timing_reset()
        for _ in range(5):
            eng.pairwise_device(leaf.data_ptr(), n, K)
        eng.synchronize()
        ms, _ = eng.timing_read(KERNEL_UNION)
        eng.timing_enable(False)
        print(f"dbg {dbg} ({names[dbg]:22s}): {ms / 5:7.3f} ms (all K2 kernels of the call)", flush=True)
