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
    """The k contraction of a (k, z) table (not-a-knot cubic spline from 504 log-k knots to the 1024 FFTLog wavenumbers, 64 redshifts x 10 000
    cosmologies = 640 000 rows) as a banded operator on the vector ALUs and, forced dense, as a GEMM on the matrix cores."""
    import torch
    from cosmoprimo_amd.spline import LinearOperator, dense_operator
    dev = torch.device('cuda', 0)
    x, xq = np.linspace(-7., 2., 504), np.linspace(-7., 2., 1024)
    banded = LinearOperator.spline(x, xq, bc='not-a-knot', device=dev)
    dense = LinearOperator.dense(dense_operator(x, xq, bc='not-a-knot'), device=dev)
    y = torch.as_tensor(np.random.default_rng(1).normal(size=(640000, 504)), device=dev)
    line = '%-62s' % '(k, z) table, k contraction: 504 -> 1024, 640 000 rows'
    for name, op, path in (('banded valu (bandwidth %d)' % banded.bandwidth, banded, None), ('dense mfma', dense, 'mfma')):
        for _ in range(5):
            out = op(y, path=path)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            out = op(y, path=path)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 5 * 1e3
        line += '  %s %.3f ms' % (name, ms)
    print(line)
    print('   (difference of the two results: %.1e)' % float((banded(y[:1000]) - dense(y[:1000], path='mfma')).abs().max()))


if __name__ == '__main__':
    main()
    spline_contraction()
