"""CPU: the N > 1 path (contiguous sharding + the final all-gather) with world_size 2 on the gloo backend."""
import os
import subprocess
import sys
import textwrap

import numpy as np

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
    n = 11                                  # odd: shards of 6 and 5 rows
    full = torch.arange(n * 3, dtype=torch.float64).reshape(n, 3)
    mine = shard(full)                      # this rank's rows
    start, stop = shard_range(n, rank, world)
    assert mine.shape[0] == stop - start and torch.equal(mine, full[start:stop])
    result = mine * 2. + 1.                 # stands for the per-row transform: no communication
    out = gather_rows(result, n_total=n)    # the one collective
    assert torch.equal(out, full * 2. + 1.), (rank, out)
    # a batch of cosmologies through a (stand-in) calculator: parameters cut per rank, results gathered, shared grids left alone
    Om, h = np.linspace(0.2, 0.4, n), np.linspace(0.6, 0.8, n)
    mine = shard_params(dict(Omega_m=Om, h=torch.as_tensor(h), n_s=0.96))
    assert mine['n_s'] == 0.96 and mine['Omega_m'].shape == (stop - start,) and torch.equal(mine['h'], torch.as_tensor(h)[start:stop])
    zgrid = np.linspace(0., 1., 4)
    local = {{'background.z': zgrid, 'background.d': np.outer(mine['Omega_m'] * mine['h'].numpy(), zgrid), 'thermodynamics.rs': mine['Omega_m'] * 100.}}
    full = gather_arrays(local, n_total=n)
    assert np.array_equal(full['background.z'], zgrid) and np.allclose(full['background.d'], np.outer(Om * h, zgrid))
    assert np.allclose(full['thermodynamics.rs'], Om * 100.)
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


def test_two_rank_gloo(tmp_path):
    script = tmp_path / 'worker.py'
    script.write_text(WORKER.format(root=ROOT))
    env = dict(os.environ, MASTER_ADDR='127.0.0.1')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1', '--master-port', '29517',
           str(script)]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stderr[-2000:]
    assert 'OK 2' in res.stdout
