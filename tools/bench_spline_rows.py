"""The FFTLog grid -> radii step of sigma_r (natural spline, 1024 geometric knots -> 256 radii, root, transposed store): the spline's tridiagonal
system solved per row in LDS (cp_spline_rows_*) against its inverse applied as a banded operator (cp_spline_apply: matrix cores / vector ALUs).
python tools/bench_spline_rows.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch      # noqa: E402
from cosmoprimo_amd.spline import LinearOperator, SplineRows      # noqa: E402

dev = torch.device('cuda', 0)
x = np.geomspace(1e-2, 1e7, 1024)
xq = np.geomspace(1., 100., 256)
banded, rows = LinearOperator.spline(x, xq, bc='natural', device=dev), SplineRows(x, xq, bc='natural', device=dev)
print('window (first knot, knots, rows per wave, halo):', rows.window)
for nb, nz in ((10000, 64), (2000, 64), (16384, 1), (4096, 1), (1024, 1)):
    y = torch.rand((nb, nz, 1024), dtype=torch.float64, device=dev) + 0.5
    line = '%6d x %2d rows:' % (nb, nz)
    for name, fn in (('banded operator', lambda: banded(y, sqrt=True, last_axis_first=True)), ('elimination in LDS', lambda: rows(y, sqrt=True, last_axis_first=True))):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            fn()
        torch.cuda.synchronize()
        line += '  %s %.3f ms' % (name, (time.perf_counter() - t0) / 10 * 1e3)
    for name, fn in (('banded, plain store', lambda: banded(y, sqrt=True)), ('elimination, plain store', lambda: rows(y, sqrt=True))):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            fn()
        torch.cuda.synchronize()
        line += '  %s %.3f ms' % (name, (time.perf_counter() - t0) / 10 * 1e3)
    print(line)
