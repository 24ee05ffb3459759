"""The FFTLog variants beside the headline one (zero padding, cropped output): log extrapolation, padded output, a size that is not half of its padded size,
a general-size transform -- 100 000 / 20 000 rows each.   python tools/bench_generic_fftlog.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch      # noqa: E402
import cosmoprimo_amd as cp      # noqa: E402

dev = torch.device('cuda', 0)


def timeit(fn):
    for _ in range(3):
        fn()
    best = 1e9
    for rnd in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            fn()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / 10 * 1e3)
    return best


for n, nrows in ((2048, 100000), (1500, 100000), (16384, 20000)):
    k = np.geomspace(1e-5, 1e2, n)
    fft = cp.PowerToCorrelation(k, device=dev)
    rows = (torch.rand((nrows, 1), dtype=torch.float64, device=dev) + 0.5) * torch.as_tensor((k / 0.05)**-1.5 / (1. + (k / 0.02)**2.2), device=dev)[None, :]
    print('%6d samples x %6d rows: extrap=0 %.3f ms | extrap=log %.3f ms | keep_padding %.3f ms' % (
        n, nrows, timeit(lambda: fft(rows)), timeit(lambda: fft(rows, extrap='log')), timeit(lambda: fft(rows, keep_padding=True))))
