"""The two kernels of config 3B timed one at a time (HIP events around 10 launches each): the library calls of one ordinary sigma_rz recorded and replayed.
    python tools/bench_config3b_kernels.py [ntables]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main(paired):
    import torch
    import cosmoprimo_amd as cp
    from cosmoprimo_amd import _lib, interpolator
    interpolator._PAIRED_TABLES = paired
    print('tables and second derivatives as %s' % ('(y, M) pairs' if paired else 'two arrays'))
    nb = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
    g = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden', 'sigma.npz'))
    rng = np.random.default_rng(1)
    amp = torch.as_tensor(rng.uniform(0.5, 2., nb), device='cuda')
    batch = amp[:, None, None] * torch.as_tensor(g['table_pk'], device='cuda')[None]
    r, zq = torch.as_tensor(g['r'], device='cuda'), torch.as_tensor(g['z'], device='cuda')
    interp = cp.PowerSpectrumInterpolator2D(g['table_k'], g['table_z'], batch)
    interp.sigma_rz(r, zq)
    lib = _lib.load()
    seen = {}
    real = {name: getattr(lib, name) for name in ('cp_tables_rows_direct', 'cp_fftlog_geospline_execute')}

    def recorder(name):
        def call(*args):
            seen[name] = args
            return real[name](*args)
        return call

    for name in real:
        setattr(lib, name, recorder(name))
    keep = interp.sigma_rz(r, zq)      # (the rows buffer is freed behind this call: a new one below)
    for name in real:
        setattr(lib, name, real[name])
    ta, ga = list(seen['cp_tables_rows_direct']), list(seen['cp_fftlog_geospline_execute'])
    rows = torch.empty((nb, ga[5], 1024), dtype=torch.float64, device='cuda')
    ta[4] = ga[2] = rows.data_ptr()
    ga[3] = keep.data_ptr()
    for name, args in (('cp_tables_rows_direct', ta), ('cp_fftlog_geospline_execute', ga)):
        ms = []
        for rep in range(3):
            for _ in range(2):
                _lib.check(real[name](*args))
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                _lib.check(real[name](*args))
            e1.record()
            torch.cuda.synchronize()
            ms.append(e0.elapsed_time(e1) / 10)
        print('%-32s %s ms' % (name, ' '.join('%.3f' % v for v in ms)))


if __name__ == '__main__':
    for paired in (False, True):
        main(paired)
