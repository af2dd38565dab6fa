"""five all-pairs calls over one realistic slab (for scripts/prof_gram.sh): python scripts/bench_gram_one.py [N] [LOG2M] [K]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dandd_amd.engine import Engine
n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
p = int(sys.argv[2]) if len(sys.argv) > 2 else 20
K = int(sys.argv[3]) if len(sys.argv) > 3 else 31
m = 1 << p
eng = Engine(0, p, True)
rng = np.random.default_rng(0)
leaf = torch.empty((n, K, m), dtype=torch.uint8, device="cuda")
for i in range(n):
    u = rng.random((K, m), dtype=np.float32) + 1e-9
    leaf[i] = torch.from_numpy(np.clip(np.floor(np.log2(300.0) - np.log2(-np.log(u))) + 1, 0, 64 - p + 1).astype(np.uint8)).cuda()
if os.environ.get("GRAM_ZEROS"):
    leaf[:, :, :: int(1 / float(os.environ["GRAM_ZEROS"]))] = int(os.environ.get("GRAM_MIN", "0"))
for _ in range(5):
    eng.pairwise_device(leaf.data_ptr(), n, K)
eng.synchronize()
