"""Config 1 (BASELINE.json: a single FFTLog of 1024 bins): latency of one call through the Python classes, numpy in/out and device tensors in/out.
python tools/latency_config1.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    import cosmoprimo_amd as cp
    k = np.logspace(-5, 2, 1024)
    pk = cp.Cosmology(engine='eisenstein_hu').get_fourier().pk_interpolator()(k, 0.)
    t0 = time.perf_counter()
    fft = cp.PowerToCorrelation(k, ell=0, lowring=True)
    fft(pk)
    torch.cuda.synchronize()
    print('setup + first call: %.2f ms' % ((time.perf_counter() - t0) * 1e3))
    tpk = torch.as_tensor(pk, device='cuda:0')
    for name, arg in [('numpy in / numpy out', pk), ('device tensor in / out', tpk)]:
        for _ in range(20):
            fft(arg)
        torch.cuda.synchronize()
        n = 500
        t0 = time.perf_counter()
        for _ in range(n):
            out = fft(arg)
        torch.cuda.synchronize()
        print('%-26s %.1f us per call' % (name, (time.perf_counter() - t0) / n * 1e6))
    from oracle import fftlog as ofl
    t = ofl.power_to_correlation(k, ell=0)
    t0 = time.perf_counter()
    for _ in range(200):
        ofl.apply(t, pk)
    print('%-26s %.1f us per call' % ('numpy port on the host', (time.perf_counter() - t0) / 200 * 1e6))


if __name__ == '__main__':
    main()
