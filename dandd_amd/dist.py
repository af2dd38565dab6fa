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


# collectives issued by this process since import, by kind (bench.py prints them: a line that claims the N>1 path
# shows how often each exchange really ran)
STATS = {"all_reduce_max_u8": 0, "all_gather": 0, "all_reduce_scalar": 0}


# The data-path collectives can go through the C ABI instead of torch.distributed (dd_allreduce_max_u8 / dd_allgather_u8:
# RCCL called by libdandd_hip.so itself, what a non-Python host binding of include/dandd_hip.h gets).  torch.distributed is
# then only the courier of the communicator's 128-byte id.
_ABI = None


def use_abi_comm(engine):
    """Open an RCCL communicator on `engine`'s context over the ranks of the live process group (rank 0 makes the id, a
    broadcast carries it) and route allreduce_max_u8 / allgather_leaves through it from now on.  Collective."""
    global _ABI
    import torch.distributed as dist
    from .engine import comm_unique_id
    rank, world = (dist.get_rank(), dist.get_world_size()) if group_live() else (0, 1)
    box = [comm_unique_id() if rank == 0 else None]
    if group_live():
        dist.broadcast_object_list(box, src=0)
    engine.comm_init(rank, world, box[0])
    _ABI = engine
    return engine


def drop_abi_comm():
    global _ABI
    if _ABI is not None:
        _ABI.comm_destroy()
    _ABI = None


def group_live():
    """A process group exists.  World size 1 counts: the collectives then still go through the backend (RCCL on a
    GPU box), which is how a one-GPU machine exercises the N>1 path's calls."""
    import torch.distributed as dist
    return dist.is_available() and dist.is_initialized()


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
    if _ABI is not None:     # RCCL behind the C ABI, on the engine's stream (the caller has set it to torch's current stream)
        assert t.is_contiguous() and t.dtype.itemsize == 1
        _ABI.allreduce_max_u8(t.data_ptr(), t.numel())
        STATS["all_reduce_max_u8"] += 1
    elif group_live():
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        STATS["all_reduce_max_u8"] += 1
    return t


def max_over_ranks(value, device=None):
    import torch
    import torch.distributed as dist
    if not group_live():
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    STATS["all_reduce_scalar"] += 1
    return float(t.item())


def gather_strings(text):
    """One short string per rank -> the list of all ranks' strings on every rank."""
    import torch.distributed as dist
    if not group_live():
        return [text]
    out = [None] * dist.get_world_size()
    dist.all_gather_object(out, text)
    return out


def sharded_ksweep(weights, K, m, sketch_into, union_into, card_of, regs=None, mine=None, device=None):
    """One pass of the sharded k-sweep on this rank -- THE N>1 path: bench.py's step and the multi-GPU tests
    both run this function (the reference has no counterpart: it parallelises over k only,
    /root/reference/lib/huffman_dandd.py:217).

    weights        sizes of ALL genomes of the job, the same list on every rank; rank r sketches
                   shard_by_weight(weights, world)[r] -- or `mine` when the caller has fixed the shard (weak
                   scaling: every rank brings its own genomes)
    sketch_into(indices, leaves)   leaves: uint8 tensor [n][K][m] <- sketches of genomes `indices`
    union_into(leaves, root)       root: uint8 tensor [K][m] <- byte-max over this rank's leaves (zeros if none)
    card_of(regs)                  float64 [(n+1)*K] cardinalities of regs = leaves followed by the root
    The only exchange is a MAX all-reduce of the root slab (RCCL ncclMax/ncclUint8 over xGMI; gloo on CPU).
    Returns (mine, regs [n+1][K][m], card [n+1][K]); row n is the root over ALL ranks' genomes."""
    import torch
    import torch.distributed as dist
    live = dist.is_available() and dist.is_initialized()
    world, rank = (dist.get_world_size(), dist.get_rank()) if live else (1, 0)
    if mine is None:
        mine = shard_by_weight(list(weights), world)[rank]
    n = len(mine)
    if regs is None:
        regs = torch.empty((n + 1, K, m), dtype=torch.uint8, device=device)
    if n:
        sketch_into(mine, regs[:n])
    union_into(regs[:n], regs[n])
    allreduce_max_u8(regs[n])
    card = np.asarray(card_of(regs), dtype=np.float64).reshape(n + 1, K)
    return mine, regs, card


def gather_rows(rows, mine, n_total, device=None):
    """Rows (float64 [len(mine)][K]) owned by the ranks -> the full [n_total][K] table on every rank
    (disjoint owners: a SUM all-reduce of the zero-padded table is a gather)."""
    import torch
    import torch.distributed as dist
    rows = np.asarray(rows, dtype=np.float64)
    full = torch.zeros((n_total, rows.shape[1] if rows.ndim == 2 else 0), dtype=torch.float64, device=device)
    if len(mine):
        full[torch.tensor(list(mine), device=device)] = torch.as_tensor(rows, dtype=torch.float64, device=device)
    if group_live():
        dist.all_reduce(full, op=dist.ReduceOp.SUM)
    return full.cpu().numpy()


def allgather_geometry(n_total, world, K, m, most_held=0):
    """Sizes of allgather_leaves' exchange, without a process group: every rank sends ONE padded [rows][K][m] uint8 slab
    (rows = the largest shard, at least ceil(n_total / world) + 1) and receives `world` of them next to the job's whole
    [n_total][K][m] slab.  -> dict(rows, sent_bytes, received_bytes, full_bytes, peak_bytes)."""
    rows = max((n_total + world - 1) // world + 1, int(most_held))
    sent = rows * K * m
    return {"rows": rows, "sent_bytes": sent, "received_bytes": world * sent, "full_bytes": n_total * K * m,
            "peak_bytes": sent + world * sent + n_total * K * m}


def allgather_leaves(leaves, mine, n_total):
    """Every rank's leaf slabs (uint8 [len(mine)][K][m], genomes `mine` of the job) -> the job's whole
    [n_total][K][m] slab on every rank, in genome order: what `progressive` and `kij` need before their
    (ordering x k) / pair schedules can be split over the ranks (ncclAllGather over xGMI --
    15 MiB for cfg 4 at log2m 14, nothing next to the sketching)."""
    import torch
    import torch.distributed as dist
    live = group_live()
    K, m = leaves.shape[1], leaves.shape[2]
    full = torch.zeros((n_total, K, m), dtype=torch.uint8, device=leaves.device)
    if not live:
        full[torch.tensor(list(mine), device=leaves.device, dtype=torch.long)] = leaves
        return full
    world = dist.get_world_size()

    def gather(t):     # every rank's `t` -> list of world tensors (torch.distributed, or dd_allgather_u8 on the tensor's bytes)
        if _ABI is None:
            out = [torch.empty_like(t) for _ in range(world)]
            dist.all_gather(out, t)
            return out
        t = t.contiguous()
        recv = torch.empty((world,) + tuple(t.shape), dtype=t.dtype, device=t.device)
        _ABI.allgather_u8(t.data_ptr(), t.numel() * t.element_size(), recv.data_ptr())
        return [recv[r] for r in range(world)]

    counts = gather(torch.tensor([len(mine)], dtype=torch.int64, device=leaves.device))
    # (shards differ by at most one genome in count when sizes are equal; padded generously: allgather_geometry)
    most = allgather_geometry(n_total, world, K, m, max(int(c.item()) for c in counts))["rows"]
    ids = torch.full((most,), -1, dtype=torch.int64, device=leaves.device)
    ids[:len(mine)] = torch.tensor(list(mine), dtype=torch.int64, device=leaves.device)
    padded = torch.zeros((most, K, m), dtype=torch.uint8, device=leaves.device)
    padded[:len(mine)] = leaves
    all_ids = gather(ids)
    all_slabs = gather(padded)
    STATS["all_gather"] += 3
    for r in range(world):
        n = int(counts[r].item())
        if n:
            full[all_ids[r][:n]] = all_slabs[r][:n]
    return full
