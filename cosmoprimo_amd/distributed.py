"""
Multi-GPU driver for the hot path: one process per GPU (``torch.distributed``, backend "nccl" = RCCL over xGMI on ROCm).

Rows (FFTLog), cosmologies (sigma(r), BAO filters) and samples (background distances) are independent, so the batch is split
into contiguous blocks, one per rank, with NO data-path collective.  The only collective is the optional final
:func:`gather_rows` (``all_gather_into_tensor``), for callers that need the full result replicated on every GPU.
On CPU the same code runs with the "gloo" backend (tests/test_distributed.py).
"""


def shard_range(n, rank, world_size):
    """Contiguous block [start, stop) of ``n`` items owned by ``rank``: sizes differ by at most one, blocks tile [0, n) in rank order."""
    if world_size < 1 or not 0 <= rank < world_size:
        raise ValueError('need 0 <= rank < world_size, got rank={}, world_size={}'.format(rank, world_size))
    base, extra = divmod(int(n), world_size)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def shard(array, rank=None, world_size=None, axis=0):
    """This rank's block of ``array`` along ``axis`` (numpy array or torch tensor); rank / world size default to the process group's."""
    if rank is None or world_size is None:
        import torch.distributed as dist
        rank, world_size = dist.get_rank(), dist.get_world_size()
    start, stop = shard_range(array.shape[axis], rank, world_size)
    index = [slice(None)] * len(array.shape)
    index[axis] = slice(start, stop)
    return array[tuple(index)]


def gather_rows(local, n_total=None, group=None):
    """
    All-gather result shards along axis 0 into the full array on every rank (the single collective of the path).

    Shards produced by :func:`shard_range` differ by at most one row: they are padded to the largest shard for
    ``all_gather_into_tensor`` and the padding is dropped.  ``n_total`` is the global number of rows (default: sum of shard sizes).
    """
    import torch
    import torch.distributed as dist
    world_size = dist.get_world_size(group)
    sizes = torch.zeros(world_size, dtype=torch.int64, device=local.device)
    sizes[dist.get_rank(group)] = local.shape[0]
    dist.all_reduce(sizes, group=group)
    sizes = [int(s) for s in sizes.tolist()]
    nmax = max(sizes)
    padded = local
    if local.shape[0] < nmax:
        padded = torch.cat([local, local.new_zeros((nmax - local.shape[0],) + tuple(local.shape[1:]))], dim=0)
    full = local.new_empty((world_size * nmax,) + tuple(local.shape[1:]))
    dist.all_gather_into_tensor(full, padded.contiguous(), group=group)
    full = full.reshape((world_size, nmax) + tuple(local.shape[1:]))
    out = torch.cat([full[r, :sizes[r]] for r in range(world_size)], dim=0)
    if n_total is not None and out.shape[0] != n_total:
        raise ValueError('gathered {} rows, expected {}'.format(out.shape[0], n_total))
    return out
