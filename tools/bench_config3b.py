"""BASELINE config 3, variant B: sigma_rz on 256 r x 64 z for a batch of tabulated P(k, z) (500 k x 30 z per cosmology;
120 000 B in + 131 072 B out per cosmology).   python tools/bench_config3b.py [ncosmo]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    import cosmoprimo_amd as cp
    nb = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
    g = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden', 'sigma.npz'))
    rng = np.random.default_rng(1)
    k, z = g['table_k'], g['table_z']
    amp = torch.as_tensor(rng.uniform(0.5, 2., nb), device='cuda')
    batch = amp[:, None, None] * torch.as_tensor(g['table_pk'], device='cuda')[None]
    r, zq = torch.as_tensor(g['r'], device='cuda'), torch.as_tensor(g['z'], device='cuda')
    t0 = time.perf_counter()
    interp = cp.PowerSpectrumInterpolator2D(k, z, batch)
    torch.cuda.synchronize()
    print('setup (sort, log-log padding, log10): %.2f ms' % ((time.perf_counter() - t0) * 1e3))
    for _ in range(6):      # (the first two calls build plans and operators: tens of milliseconds)
        interp.sigma_rz(r, zq)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        out = interp.sigma_rz(r, zq)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
    nbytes = nb * (500 * 30 * 8 + 256 * 64 * 8)
    print('config 3B: sigma_rz 256 r x 64 z of %d tabulated P(k, z): %.2f ms, %.3g cosmologies/s, %.1f GB/s algorithmic (%.2f %% of 8 TB/s)' % (
        nb, dt * 1e3, nb / dt, nbytes / dt / 1e9, nbytes / dt / 8e12 * 100))


if __name__ == '__main__':
    main()
