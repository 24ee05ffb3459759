"""Padded sizes beyond the fused kernel (Np > 8192): time per transform and algorithmic bandwidth of the large-size path.
python tools/bench_fftlog_large.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    import cosmoprimo_amd as cp
    dev = torch.device('cuda', 0)
    for n in (4096, 8192, 32768, 262144, 2097152):
        k = np.logspace(-4, 2, n)
        f = cp.PowerToCorrelation(k, ell=0)
        nrows = max((1 << 26) // n, 2)
        fun = torch.rand((nrows, n), dtype=torch.float64, device=dev) + 0.5
        for _ in range(3):
            out = f(fun)[1]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        nrep = 10
        for _ in range(nrep):
            out = f(fun)[1]
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / nrep
        print('n = %8d (Np = %8d) x %6d rows: %8.3f ms, %.3g transforms/s, %.0f GB/s algorithmic (16 n B per row)' % (
            n, f.padded_size, nrows, dt * 1e3, nrows / dt, 16. * n * nrows / dt / 1e9))


if __name__ == '__main__':
    main()
