"""Batched many-matrices driver: the only way this path uses several GPUs.

A single factorisation is one dependency chain and stays on one GPU.  Independent matrices are dealt
round-robin over the ranks of a `torch.distributed` job (one process per GPU, backend "nccl" = RCCL on
ROCm, or "gloo" on CPU for tests); each rank runs the unmodified single-GPU path; there is NO collective
on the data path -- only one all-gather of small per-matrix records at the end.
"""


def shard(n_items, rank, world):
    """indices of the batch members rank `rank` of `world` owns (round-robin)"""
    return list(range(rank, n_items, world))


def run_batch(n_items, work, rank=None, world=None, group=None):
    """`work(i)` -> picklable record for matrix i (e.g. (nnzL, nnzU, checksum, seconds)).
    Returns the list of all n_items records, in matrix order, on every rank."""
    import torch.distributed as dist
    distributed = dist.is_available() and dist.is_initialized()
    if rank is None:
        rank = dist.get_rank(group) if distributed else 0
    if world is None:
        world = dist.get_world_size(group) if distributed else 1
    mine = [(i, work(i)) for i in shard(n_items, rank, world)]
    if not distributed or world == 1:
        parts = [mine]
    else:
        parts = [None] * world
        dist.all_gather_object(parts, mine, group=group)
    out = [None] * n_items
    for part in parts:
        for i, rec in part:
            out[i] = rec
    return out
