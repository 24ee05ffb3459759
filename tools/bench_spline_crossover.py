"""Where the FFTLog + spline kernels cross: the band operator inside the transform's kernel (cp_fftlog_spline_execute, rows 2 ... 8192 today) against the
prefiltered B-spline form (cp_fftlog_geospline_execute, 8193 rows and more), 256 radii, for batches in between.   python tools/bench_spline_crossover.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch      # noqa: E402
import cosmoprimo_amd as cp      # noqa: E402
from cosmoprimo_amd import interpolator as itp      # noqa: E402
from cosmoprimo_amd.spline import LinearOperator      # noqa: E402

dev = torch.device('cuda', 0)
k = np.geomspace(1e-7, 1e2, 1024)
fft = cp.TophatVariance(k, device=dev)
s, r = fft.y[0], np.geomspace(1., 100., 256)
op = LinearOperator.spline(s, r, bc='natural', device=dev)
base = torch.as_tensor((k / 0.05)**-1.5 * 1e3 / (1. + (k / 0.02)**2.2), device=dev)
for nrows in (64, 256, 1024, 2048, 4096, 8192, 16384):
    rows = (torch.rand((nrows, 1), dtype=torch.float64, device=dev) + 0.5) * base[None, :]
    out = {}
    for name, fn in (('band operator in the kernel', lambda: itp._fftlog_then_spline(fft, op, rows, dev, sqrt=True)),
                     ('prefiltered B-spline', lambda: itp._fftlog_then_geospline(fft, s, r, rows, dev, sqrt=True))):
        if fn() is None:
            out[name] = float('nan')
            continue
        best = 1e9
        for rnd in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(50):
                fn()
            torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t0) / 50 * 1e3)
        out[name] = best
    print('%6d rows: %s' % (nrows, ' | '.join('%s %.4f ms' % kv for kv in out.items())))
