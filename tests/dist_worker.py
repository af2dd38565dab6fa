"""Worker of tests/test_distributed.py: one rank of a gloo world; the GPU calls of
dandd_amd.dist.distributed_ksweep are replaced by the CPU oracle (checker) so the sharding +
all-reduce logic runs without a GPU."""
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dandd_amd import dist as dd  # noqa: E402
from oracle import dd_oracle as orc  # noqa: E402


def main():
    out_path, kmin, kmax, p = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    fastas = sys.argv[5:]
    rank, _, world = dd.env_ranks()
    dist.init_process_group("gloo", rank=rank, world_size=world)
    m = 1 << p
    sizes = [os.path.getsize(f) for f in fastas]
    touched = []

    def sketch_fn(path):
        touched.append(os.path.basename(path))
        return torch.from_numpy(orc.sketch_sweep(np.fromfile(path, dtype=np.uint8), kmin, kmax, p))

    def union_fn(slabs):
        return torch.from_numpy(orc.union(*[s.numpy() for s in slabs]))

    def card_fn(regs):
        r = regs.numpy()
        return np.array([orc.card(r[i], p) for i in range(r.shape[0])])

    leaf_card, root, root_card = dd.distributed_ksweep(fastas, sizes, kmin, kmax, m, sketch_fn, union_fn, card_fn)
    slowest = dd.max_over_ranks(float(rank + 1))
    with open(f"{out_path}.{rank}", "w") as f:
        json.dump({"rank": rank, "world": world, "touched": touched, "leaf_card": leaf_card.tolist(),
                   "root_sha": int(np.frombuffer(root.numpy().tobytes(), dtype=np.uint8).astype(np.uint64).sum()),
                   "root_card": root_card.tolist(), "slowest": slowest}, f)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
