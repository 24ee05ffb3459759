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


def gather_rows(local, n_total=None, group=None, out=None):
    """
    All-gather result shards along axis 0 into the full array on every rank (the single collective of the path).

    ``n_total`` is the global number of rows; the shards are then known to be the blocks of :func:`shard_range` and no size exchange
    is needed (without it the sizes are exchanged first).  Equal shards -- the benchmark's case -- take ONE ``all_gather_into_tensor``
    straight into the result (``out`` if given: a preallocated ``(n_total, ...)`` tensor, e.g. reused from call to call); unequal shards
    differ by at most one row and go through a buffer padded to the largest shard, from which the valid rows are copied.
    """
    import torch
    import torch.distributed as dist
    world_size, rank = dist.get_world_size(group), dist.get_rank(group)
    if n_total is None:
        sizes = torch.zeros(world_size, dtype=torch.int64, device=local.device)
        sizes[rank] = local.shape[0]
        dist.all_reduce(sizes, group=group)
        sizes = [int(s) for s in sizes.tolist()]
        n_total = sum(sizes)
    else:
        sizes = [b - a for a, b in (shard_range(n_total, r, world_size) for r in range(world_size))]
        if local.shape[0] != sizes[rank]:
            raise ValueError('rank {} holds {} rows, its block of {} rows over {} ranks has {}'.format(rank, local.shape[0], n_total, world_size, sizes[rank]))
    tail = tuple(local.shape[1:])
    if out is None:
        out = local.new_empty((n_total,) + tail)
    elif tuple(out.shape) != (n_total,) + tail or out.dtype != local.dtype or not out.is_contiguous():
        raise ValueError('out must be a contiguous {} tensor of shape {}'.format(local.dtype, (n_total,) + tail))
    if n_total == 0:
        return out
    nmax = max(sizes)
    if min(sizes) == nmax:
        dist.all_gather_into_tensor(out, local.contiguous(), group=group)
        return out
    padded = local.new_zeros((nmax,) + tail)
    padded[:local.shape[0]] = local
    buffer = local.new_empty((world_size, nmax) + tail)
    dist.all_gather_into_tensor(buffer.reshape((world_size * nmax,) + tail), padded, group=group)
    start = 0
    for r, size in enumerate(sizes):
        out[start:start + size] = buffer[r, :size]
        start += size
    return out


def shard_params(params, rank=None, world_size=None):
    """
    This rank's block of a batch of cosmologies: every array-valued parameter (numpy array or torch tensor with one entry per cosmology) is
    cut to the rank's contiguous block, scalars and other values are passed through.  For the batch drivers::

        calculator = get_calculator(Cosmology(engine='eisenstein_hu'))
        mine = calculator(**shard_params(dict(Omega_m=Omega_m, h=h)))        # no communication
        full = gather_arrays(mine, n_total=Omega_m.size)                     # optional: replicate the results
    """
    sizes = {int(v.shape[0]) for v in params.values() if hasattr(v, 'shape') and len(v.shape) >= 1}
    if len(sizes) > 1:
        raise ValueError('array-valued parameters must share one length, got {}'.format(sorted(sizes)))
    if not sizes:
        return dict(params)
    return {name: shard(v, rank=rank, world_size=world_size) if hasattr(v, 'shape') and len(v.shape) >= 1 else v for name, v in params.items()}


# last component of the names of result entries that are grids shared by the whole batch (the calculators' 'background.z', 'fourier.k', ...)
SHARED_GRID_NAMES = ('z', 'k', 's', 'r', 'ell')


def gather_arrays(local, n_total, batch_keys=None, shared_keys=None, device=None, group=None):
    """
    All-gather a dictionary of result arrays (as returned by a calculator on this rank's block) along the batch axis.

    ``batch_keys`` names the entries that carry the batch axis.  By default they are the entries whose leading dimension is the size of
    the owner's block ON EVERY RANK (the dimensions are exchanged first, so that all ranks take the same decision -- a rank must never
    infer it from its own block: with uneven blocks the ranks would disagree on the list of collectives) and that are not shared grids:
    ``shared_keys``, by default the entries named ``*.z``, ``*.k``, ``*.s``, ``*.r``, ``*.ell`` (a block of 256 cosmologies has the
    length of a 256-point redshift grid; only the name tells them apart).  Shared entries are returned as they are.
    numpy in, numpy out: arrays travel through ``device`` (default: CPU for gloo, the current GPU for nccl / RCCL).
    """
    import numpy as np
    import torch
    import torch.distributed as dist
    if device is None:
        device = torch.device('cuda', torch.cuda.current_device()) if dist.get_backend(group) == 'nccl' else torch.device('cpu')
    world_size = dist.get_world_size(group)
    names = sorted(local)
    if batch_keys is None:
        if shared_keys is None:
            shared_keys = [name for name in names if name.split('.')[-1] in SHARED_GRID_NAMES]
        dims = torch.tensor([np.shape(local[name])[0] if np.ndim(local[name]) >= 1 else -1 for name in names], dtype=torch.int64, device=device)
        all_dims = torch.empty((world_size, len(names)), dtype=torch.int64, device=device)
        dist.all_gather_into_tensor(all_dims.reshape(-1), dims, group=group)      # fails loudly if the ranks hold different entries
        blocks = torch.tensor([b - a for a, b in (shard_range(n_total, r, world_size) for r in range(world_size))], dtype=torch.int64, device=device)
        is_batch = (all_dims == blocks[:, None]).all(dim=0).cpu().tolist()
        batch_keys = [name for name, flag in zip(names, is_batch) if flag and name not in shared_keys]
    out = dict(local)
    for name in sorted(batch_keys):      # same order on every rank
        t = torch.as_tensor(np.ascontiguousarray(local[name])).to(device)
        out[name] = gather_rows(t, n_total=n_total, group=group).cpu().numpy()
    return out
