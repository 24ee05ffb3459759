"""Config 3 (sigma_rz of 10 000 EH98 cosmologies at 256 r x 64 z, one fused kernel): start offsets between the workgroups of a CU, so that they do
not evaluate together and store together.  CP_SIGMA_STAGGER = "div,mod,sleeps": workgroup b waits ((b / div) % mod) x sleeps x s_sleep(127) before its
first pair.  python tools/ab_sigma_stagger.py"""
import os
import sys
import time
import warnings

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main(ncosmo=10000, reps=20):
    import torch
    import cosmoprimo_amd as cp
    import bench
    warnings.simplefilter('ignore')
    dev = torch.device('cuda:0')
    cosmo = cp.Cosmology(engine='eisenstein_hu', sigma8=0.8, **bench.eh_parameters(ncosmo, 1, torch, dev))
    interp = cosmo.get_fourier().pk_interpolator()
    r, z = torch.as_tensor(np.geomspace(1, 100, 256), device=dev), torch.as_tensor(np.linspace(0, 3, 64), device=dev)
    os.environ.pop('CP_SIGMA_STAGGER', None)
    ref = interp.sigma_rz(r, z)

    def timed():
        for _ in range(5):
            interp.sigma_rz(r, z)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(reps):
            out = interp.sigma_rz(r, z)
        torch.cuda.synchronize(dev)
        return (time.perf_counter() - t0) / reps * 1e3, out

    settings = [None] + ['%d,%d,%d' % (d, m, s) for d, m in [(1, 2), (1, 4), (8, 4), (256, 4), (256, 2), (32, 4), (1, 8)] for s in (1, 2, 4, 6, 9)] + [None]
    for setting in settings:
        if setting is None:
            os.environ.pop('CP_SIGMA_STAGGER', None)
        else:
            os.environ['CP_SIGMA_STAGGER'] = setting
        ms, out = timed()
        print('CP_SIGMA_STAGGER=%-12s %.4f ms   same bits: %s' % (setting, ms, bool(torch.equal(out, ref))))


if __name__ == '__main__':
    main()
