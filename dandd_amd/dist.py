"""Multi-GPU sharding of the sketching path: one process per GPU, `torch.distributed` (backend
"nccl" = RCCL over xGMI on ROCm; "gloo" in the CPU tests).

The path shards with no data-path collective: (genome x k-sweep) sketch jobs are independent
(the reference itself only ever parallelises over k, lib/huffman_dandd.py:217).
The single exchange is the root union: every rank holds the byte-max union of ITS genomes'
[K][m] register slabs and an all-reduce with MAX over uint8 produces the union over all genomes
(592 KiB at log2m=14, K=37).  Per-genome cardinalities are a tiny all-gather.
"""
import os

import numpy as np


def env_ranks():
    """(rank, local_rank, world_size) from the torch.distributed.run environment."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def shard_by_weight(weights, world):
    """Longest-processing-time assignment of items to ranks; returns a list of index lists.

    Deterministic (ties broken by index), so every rank computes the same plan without talking."""
    order = sorted(range(len(weights)), key=lambda i: (-weights[i], i))
    loads = [0] * world
    plan = [[] for _ in range(world)]
    for i in order:
        r = min(range(world), key=lambda q: (loads[q], q))
        plan[r].append(i)
        loads[r] += weights[i]
    for p in plan:
        p.sort()
    return plan


def allreduce_max_u8(t):
    """In-place MAX all-reduce of a uint8 tensor (the HLL union across ranks)."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return t


def max_over_ranks(value, device=None):
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def distributed_ksweep(fastas, sizes, kmin, kmax, m, sketch_fn, union_fn, card_fn, device=None):
    """Sketch `fastas` over k in [kmin, kmax], sharded by size over the ranks of the default group.

    sketch_fn(path) -> uint8 [K][m] (torch tensor on `device`)   this rank's leaf sketch
    union_fn(list of [K][m]) -> [K][m]                           local N-way byte max
    card_fn([J][m]) -> float64 ndarray [J]                       cardinalities
    Returns on every rank: (leaf_card [n][K], root [K][m] tensor, root_card [K]).
    """
    import torch
    import torch.distributed as dist
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    K, n = kmax - kmin + 1, len(fastas)
    mine = shard_by_weight(list(sizes), world)[rank]
    slabs = [sketch_fn(fastas[i]) for i in mine]
    leaf_card = torch.zeros((n, K), dtype=torch.float64, device=device)
    if slabs:
        cards = card_fn(torch.stack(slabs).reshape(-1, m)).reshape(len(slabs), K)
        leaf_card[torch.tensor(mine, device=device)] = torch.as_tensor(cards, dtype=torch.float64, device=device)
        root = union_fn(slabs)
    else:
        root = torch.zeros((K, m), dtype=torch.uint8, device=device)
    if world > 1:
        dist.all_reduce(leaf_card, op=dist.ReduceOp.SUM)  # disjoint rows: sum == gather
        allreduce_max_u8(root)
    root_card = np.asarray(card_fn(root.reshape(-1, m)), dtype=np.float64)
    return leaf_card.cpu().numpy(), root, root_card
