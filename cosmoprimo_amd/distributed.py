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


def shard_params(params, rank=None, world_size=None):
    """
    This rank's block of a batch of cosmologies: every array-valued parameter (numpy array or torch tensor with one entry per cosmology) is
    cut to the rank's contiguous block, scalars and other values are passed through.  For the batch drivers::

        calculator = get_calculator(Cosmology(engine='eisenstein_hu'))
        mine = calculator(**shard_params(dict(Omega_m=Omega_m, h=h)))        # no communication
        full = gather_arrays(mine, n_total=Omega_m.size)                     # optional: replicate the results
    """
    import numpy as np
    sizes = {int(v.shape[0]) for v in params.values() if hasattr(v, 'shape') and len(v.shape) >= 1}
    if len(sizes) > 1:
        raise ValueError('array-valued parameters must share one length, got {}'.format(sorted(sizes)))
    if not sizes:
        return dict(params)
    return {name: shard(v, rank=rank, world_size=world_size) if hasattr(v, 'shape') and len(v.shape) >= 1 else v for name, v in params.items()}


def gather_arrays(local, n_total=None, batch_keys=None, device=None, group=None):
    """
    All-gather a dictionary of result arrays (as returned by a calculator on this rank's block) along the batch axis.  ``batch_keys``: the
    entries that carry the batch axis (default: those whose leading dimension equals this rank's block size; grids shared by the batch such
    as 'fourier.k' are returned as they are).  numpy in, numpy out: arrays travel through ``device`` (default: CPU for gloo, the current GPU
    for nccl / RCCL).
    """
    import numpy as np
    import torch
    import torch.distributed as dist
    if device is None:
        device = torch.device('cuda', torch.cuda.current_device()) if dist.get_backend(group) == 'nccl' else torch.device('cpu')
    if batch_keys is None:
        nloc = {int(np.shape(v)[0]) for v in local.values() if np.ndim(v) >= 1}
        if n_total is None:
            raise ValueError('give n_total or batch_keys')
        start, stop = shard_range(n_total, dist.get_rank(group), dist.get_world_size(group))
        batch_keys = [name for name, v in local.items() if np.ndim(v) >= 1 and np.shape(v)[0] == stop - start]
    out = dict(local)
    for name in sorted(batch_keys):      # same order on every rank
        t = torch.as_tensor(np.ascontiguousarray(local[name])).to(device)
        out[name] = gather_rows(t, n_total=n_total, group=group).cpu().numpy()
    return out
