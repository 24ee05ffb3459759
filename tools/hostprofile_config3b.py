"""Host-side profile (cProfile) of config 3, variant B: where the Python time of one sigma_rz call on 10 000 tables goes.  python tools/hostprofile_config3b.py"""
import cProfile
import os
import pstats
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    import cosmoprimo_amd as cp
    nb = 10000
    g = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden', 'sigma.npz'))
    rng = np.random.default_rng(1)
    k, z = g['table_k'], g['table_z']
    amp = torch.as_tensor(rng.uniform(0.5, 2., nb), device='cuda')
    batch = amp[:, None, None] * torch.as_tensor(g['table_pk'], device='cuda')[None]
    r, zq = torch.as_tensor(g['r'], device='cuda'), torch.as_tensor(g['z'], device='cuda')
    interp = cp.PowerSpectrumInterpolator2D(k, z, batch)
    for _ in range(3):
        interp.sigma_rz(r, zq)
    torch.cuda.synchronize()
    prof = cProfile.Profile()
    t0 = time.perf_counter()
    prof.enable()
    for _ in range(4):
        interp.sigma_rz(r, zq)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    prof.disable()
    print('host returns after %.2f ms per call, device done after %.2f ms per call' % ((t1 - t0) / 4 * 1e3, (time.perf_counter() - t0) / 4 * 1e3))
    st = pstats.Stats(prof)
    st.sort_stats('cumulative').print_stats(30)
    st.sort_stats('tottime').print_stats(12)


if __name__ == '__main__':
    main()
