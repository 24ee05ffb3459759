"""FFTLog rows beyond the LDS-resident kernel (Np > 8192: cp_fftlog_large.hip): rows/s and algorithmic GB/s (16 N bytes per row).
    python tools/bench_fftlog_large.py [N] [nrows]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    import cosmoprimo_amd as cp
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
    nrows = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
    k = np.geomspace(1e-5, 1e3, n)
    fft = cp.PowerToCorrelation(k, ell=0, q=0)
    rows = torch.rand((nrows, n), dtype=torch.float64, device='cuda') + 0.5
    for _ in range(3):
        s, out = fft(rows)
    torch.cuda.synchronize()
    ms = []
    for rep in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            s, out = fft(rows)
        e1.record()
        torch.cuda.synchronize()
        ms.append(e0.elapsed_time(e1) / 5)
    best = min(ms)
    print('N %d (padded %d), %d rows: %s ms -> %.3e rows/s, %.0f GB/s algorithmic' % (n, fft.padded_size, nrows, ' '.join('%.3f' % v for v in ms), nrows / best * 1e3,
                                                                                       16. * n * nrows / best / 1e6))


if __name__ == '__main__':
    main()
