"""CPU: the N > 1 path (contiguous sharding + the final all-gather) with world sizes 2 and 3 on the gloo backend."""
import os
import subprocess
import sys
import textwrap

import numpy as np
import pytest

from conftest import ROOT
from cosmoprimo_amd.distributed import shard_range


def test_shard_range_tiles():
    for n in [0, 1, 7, 8, 100000, 100001]:
        for world in [1, 2, 3, 8]:
            blocks = [shard_range(n, r, world) for r in range(world)]
            assert blocks[0][0] == 0 and blocks[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(blocks[:-1], blocks[1:]))
            sizes = [b - a for a, b in blocks]
            assert max(sizes) - min(sizes) <= 1


WORKER = textwrap.dedent('''
    import os, sys
    import numpy as np, torch, torch.distributed as dist
    sys.path.insert(0, {root!r})
    from cosmoprimo_amd.distributed import shard, shard_range, gather_rows, shard_params, gather_arrays
    dist.init_process_group('gloo')
    rank, world = dist.get_rank(), dist.get_world_size()

    def transform(rows):                    # stands for the per-row kernel (row-wise, no communication): a cumulative sum along the row
        return torch.cumsum(rows, dim=-1) * 2. + 1.

    # rows: uneven shards (n not a multiple of the world size), equal shards, fewer rows than ranks, nothing at all
    for n in (11, 4 * world, world - 1, 0):
        full = torch.arange(n * 3, dtype=torch.float64).reshape(n, 3)
        mine = shard(full)
        start, stop = shard_range(n, rank, world)
        assert mine.shape[0] == stop - start and torch.equal(mine, full[start:stop])
        out = gather_rows(transform(mine), n_total=n)                     # the one collective
        assert torch.equal(out, transform(full)), (rank, n, out)
        assert torch.equal(gather_rows(transform(mine)), transform(full))   # sizes exchanged when n_total is not given
        buf = torch.full((n, 3), -1., dtype=torch.float64)
        assert gather_rows(transform(mine), n_total=n, out=buf) is buf and torch.equal(buf, transform(full))
    try:
        gather_rows(torch.zeros(5, 3, dtype=torch.float64), n_total=3 * world)
        raise SystemExit('no error for a shard that is not the rank block')
    except ValueError:
        pass

    # a batch of cosmologies through a (stand-in) calculator: parameters cut per rank, results gathered, shared grids left alone
    def calculator(Omega_m, h, n_s, nz):
        zgrid = np.linspace(0., 1., nz)
        h = h.numpy() if hasattr(h, 'numpy') else h
        return {{'background.z': zgrid, 'background.comoving_radial_distance': np.outer(Omega_m * h, zgrid) + n_s, 'thermodynamics.rs_drag': Omega_m * 100.,
                 'fourier.k': np.logspace(-3, 0, 5), 'fourier.pk': np.multiply.outer(Omega_m, np.ones((5, nz)))}}

    for n, nz in ((11, 4), (7, 4), (8 * world, 8), (3 * world + 1, 3)):     # (7, 4) with 2 ranks: rank 0 holds 4 rows, the size of the z grid;
        Om, h = np.linspace(0.2, 0.4, n), np.linspace(0.6, 0.8, n)        # (8 world, 8): every block has the size of the z grid
        mine = shard_params(dict(Omega_m=Om, h=torch.as_tensor(h), n_s=0.96))
        start, stop = shard_range(n, rank, world)
        assert mine['n_s'] == 0.96 and mine['Omega_m'].shape == (stop - start,) and torch.equal(mine['h'], torch.as_tensor(h)[start:stop])
        local = calculator(nz=nz, **mine)
        full = gather_arrays(local, n_total=n)
        want = calculator(Om, h, 0.96, nz)
        assert sorted(full) == sorted(want)
        for name in want:
            assert full[name].shape == want[name].shape and np.allclose(full[name], want[name]), (n, nz, name, full[name].shape)
        named = gather_arrays(local, n_total=n, batch_keys=['thermodynamics.rs_drag'])
        assert named['thermodynamics.rs_drag'].shape == (n,) and named['fourier.pk'].shape == local['fourier.pk'].shape
    try:
        shard_params(dict(a=np.zeros(3), b=np.zeros(4)))
        raise SystemExit('no error for ragged parameters')
    except ValueError:
        pass
    dist.barrier()
    if rank == 0:
        print('OK', world)
    dist.destroy_process_group()
''')


@pytest.mark.parametrize('world', [2, 3])
def test_gloo_ranks(tmp_path, world):
    script = tmp_path / 'worker.py'
    script.write_text(WORKER.format(root=ROOT))
    env = dict(os.environ, MASTER_ADDR='127.0.0.1')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(world), '--master-addr', '127.0.0.1', '--master-port',
           str(29517 + world), str(script)]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stderr[-3000:]
    assert 'OK %d' % world in res.stdout
