"""Dense operators: the matrix-core (MFMA f64) kernel against the banded vector-ALU kernel, for the shapes the package uses.
    python tools/bench_linop.py            (under rocprofv3 --kernel-trace --stats for the committed comparison)"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    from cosmoprimo_amd.spline import LinearOperator
    dev = torch.device('cuda', 0)
    rng = np.random.default_rng(0)
    cases = [('kirkby2013 / f2 filters: 1024 x 1024, 16 384 columns', 1024, 1024, 16384),
             ('brieden2022 envelopes: 341 x 341, 16 384 columns', 341, 341, 16384),
             ('Simpson sigma_r: 1024 nodes -> 256 radii, 10 000 spectra', 1024, 256, 10000),
             ('(k, z) table, z contraction: 30 -> 64 redshifts, 640 000 rows', 30, 64, 640000)]
    for name, n, nq, nrows in cases:
        op = LinearOperator.dense(rng.normal(size=(nq, n)), device=dev)
        y = torch.as_tensor(rng.normal(size=(nrows, n)), device=dev)
        line = '%-62s' % name
        for path in ('valu', 'mfma'):
            for _ in range(20):
                op(y, path=path)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(10):
                op(y, path=path)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / 10 * 1e3
            line += '  %s %.3f ms (%.1f TFLOP/s)' % (path, ms, 2. * n * nq * nrows / ms / 1e9)
        print(line)


def spline_contraction():
    """Spline operators of the package as banded operators on the vector ALUs and as block-banded GEMMs on the matrix cores (a tile of 64 queries
    times the window of knots under its bands): the k contraction of (k, z) tables, the FFTLog grid to radii, the two operators of wallish2018."""
    import torch
    from cosmoprimo_amd.spline import LinearOperator
    dev = torch.device('cuda', 0)
    klin = np.linspace(1e-4, 10., 3666)
    cases = [('(k, z) tables, k contraction: 504 -> 1024 log k, 640 000 rows', np.linspace(-7., 2., 504), np.linspace(-7., 2., 1024), 'not-a-knot', 0, 640000),
             ('FFTLog grid -> radii: 1024 -> 256, 640 000 rows', np.linspace(-2., 7., 1024), np.linspace(0., 2., 256), 'natural', 0, 640000),
             ('wallish2018 second derivatives: 2048 -> 2048, 32 768 rows', 1. + np.arange(2048.), 1. + np.arange(2048.), 'clamped', 2, 32768),
             ('wallish2018 splice: 3666 linear knots -> 1024 log k, 16 384 rows', klin, np.geomspace(2e-4, 9.9, 1024), 'clamped', 0, 16384)]
    for name, x, xq, bc, nu, nrows in cases:
        op = LinearOperator.spline(x, xq, bc=bc, nu=nu, device=dev)
        y = torch.as_tensor(np.random.default_rng(1).normal(size=(nrows, x.size)), device=dev)
        line = '%-68s bandwidth %3d' % (name, op.bandwidth)
        for path in ('valu', 'mfma', None):
            for _ in range(5):
                op(y, path=path)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(5):
                op(y, path=path)
            torch.cuda.synchronize()
            line += '  %s %.3f ms' % (path or 'default', (time.perf_counter() - t0) / 5 * 1e3)
        print(line + '  (paths differ by %.1e)' % float((op(y[:1000], path='valu') - op(y[:1000], path='mfma')).abs().max()))


if __name__ == '__main__':
    main()
    spline_contraction()
