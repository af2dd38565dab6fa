"""Worker of tests/test_distributed.py: one rank of a gloo world; the GPU calls of
dandd_amd.dist.sharded_ksweep (the function bench.py steps through) are replaced by the CPU oracle (checker) so the sharding +
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

    K = kmax - kmin + 1

    def sketch_into(indices, leaves):
        for j, i in enumerate(indices):
            touched.append(os.path.basename(fastas[i]))
            leaves[j] = torch.from_numpy(orc.sketch_sweep(np.fromfile(fastas[i], dtype=np.uint8), kmin, kmax, p))

    def union_into(leaves, root):
        root.zero_()
        for j in range(leaves.shape[0]):
            torch.maximum(root, leaves[j], out=root)

    def card_of(regs):
        r = regs.reshape(-1, m).numpy()
        return np.array([orc.card(r[i], p) for i in range(r.shape[0])])

    mine, regs, card = dd.sharded_ksweep(sizes, K, m, sketch_into, union_into, card_of)
    leaf_card = dd.gather_rows(card[:len(mine)], mine, len(fastas))
    root, root_card = regs[len(mine)], card[len(mine)]
    gathered = dd.allgather_leaves(regs[:len(mine)], mine, len(fastas))   # the whole job's leaf slab on every rank
    slowest = dd.max_over_ranks(float(rank + 1))
    with open(f"{out_path}.{rank}", "w") as f:
        json.dump({"rank": rank, "world": world, "touched": touched, "leaf_card": leaf_card.tolist(),
                   "root_sha": int(np.frombuffer(root.numpy().tobytes(), dtype=np.uint8).astype(np.uint64).sum()),
                   "root_card": root_card.tolist(), "slowest": slowest,
                   "gathered_sums": [int(gathered[i].to(torch.int64).sum()) for i in range(len(fastas))]}, f)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
