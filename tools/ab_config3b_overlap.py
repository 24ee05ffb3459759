"""Config 3B (sigma_rz of 10 000 tabulated P(k, z)): its two kernels one after the other (the shipped route) against the same two kernels over chunks of
tables on two streams, the tables kernel of chunk i+1 (memory-bound) beside the FFTLog kernel of chunk i (issue-bound).

    python tools/ab_config3b_overlap.py [chunks ...]

The library calls of the shipped route are recorded once (handles, pointers) and re-issued per chunk with offset pointers.
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main(chunk_counts, ncosmo=10000, reps=10):
    import torch
    import cosmoprimo_amd as cp
    from cosmoprimo_amd import _lib

    dev = torch.device('cuda:0')
    g = np.load(os.path.join(ROOT, 'tests', 'golden', 'sigma.npz'))
    rng = np.random.default_rng(1)
    amp = torch.as_tensor(rng.uniform(0.5, 2., ncosmo), device=dev)
    tables = amp[:, None, None] * torch.as_tensor(g['table_pk'], device=dev)[None]
    interp = cp.PowerSpectrumInterpolator2D(g['table_k'], g['table_z'], tables)
    r, z = torch.as_tensor(np.geomspace(1, 100, 256), device=dev), torch.as_tensor(np.linspace(0, 3, 64), device=dev)
    ref = interp.sigma_rz(r, z)
    torch.cuda.synchronize(dev)

    lib = _lib.load()
    calls = {}
    originals = {name: getattr(lib, name) for name in ('cp_tables_rows_direct', 'cp_fftlog_geospline_execute')}

    def recorder(name):
        def call(*args):
            calls[name] = args
            return originals[name](*args)
        return call

    for name in originals:
        setattr(lib, name, recorder(name))
    interp.sigma_rz(r, z)
    torch.cuda.synchronize(dev)
    for name, fn in originals.items():
        setattr(lib, name, fn)
    kplan, opy, m_ptr, _, _, nb, mode, scale, _ = calls['cp_tables_rows_direct']
    fplan, gplan, _, _, nrows, group, sqrt, _ = calls['cp_fftlog_geospline_execute']
    assert nb == ncosmo and nrows == ncosmo * 64 and group == 64, (nb, nrows, group)
    nz, nk, nr = 64, 1024, 256
    _, nz_in, nk_in, _ = interp._interp._fun_y_major_m.shape       # the tables with the knots of the extrapolation on either side
    print('tables as the kernel reads them:', tuple(interp._interp._fun_y_major_m.shape))
    rows = torch.empty((ncosmo, nz, nk), dtype=torch.float64, device=dev)
    out = torch.empty((ncosmo, nr, nz), dtype=torch.float64, device=dev)

    def timed(fn):
        for _ in range(3):
            fn()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize(dev)
        return (time.perf_counter() - t0) / reps * 1e3

    print('shipped route through sigma_rz: %.3f ms' % timed(lambda: interp.sigma_rz(r, z)))
    main_stream = torch.cuda.current_stream(dev)
    side = torch.cuda.Stream(dev)

    def run(nchunks, side=side):
        bounds = np.linspace(0, ncosmo, nchunks + 1).astype(int)
        done = [torch.cuda.Event() for _ in range(nchunks)]
        side.wait_stream(main_stream)
        for i in range(nchunks):
            lo, n = int(bounds[i]), int(bounds[i + 1] - bounds[i])
            _lib.check(lib.cp_tables_rows_direct(kplan, opy, m_ptr + lo * nz_in * nk_in * 16, None, rows.data_ptr() + lo * nz * nk * 8, n, mode, scale,
                                                 side.cuda_stream))
            done[i].record(side)
        for i in range(nchunks):
            lo, n = int(bounds[i]), int(bounds[i + 1] - bounds[i])
            main_stream.wait_event(done[i])
            _lib.check(lib.cp_fftlog_geospline_execute(fplan, gplan, rows.data_ptr() + lo * nz * nk * 8, out.data_ptr() + lo * nr * nz * 8, n * nz, group, sqrt,
                                                       main_stream.cuda_stream))

    run(1)
    torch.cuda.synchronize(dev)
    rows_ref = rows.clone()
    for nchunks in chunk_counts:
        ms = timed(lambda: run(nchunks))
        same = bool(torch.equal(out, ref))
        print('    rows of the tables kernel the same: %s' % bool(torch.equal(rows, rows_ref)))
        bad = (out != ref).flatten(1).any(dim=1)
        print('%3d chunk(s) on two streams: %.3f ms   same bits as the shipped route: %s (%d tables differ, first %s, max rel %.3g)'
              % (nchunks, ms, same, int(bad.sum()), bad.nonzero()[:3].flatten().tolist(), float(((out - ref).abs() / ref.abs()).max())))
        ms = timed(lambda: run(nchunks, side=main_stream))
        print('%3d chunk(s) on one stream:  %.3f ms   same bits: %s' % (nchunks, ms, bool(torch.equal(out, ref))))


if __name__ == '__main__':
    main([int(a) for a in sys.argv[1:]] or [1, 2, 4, 8, 16, 32])
