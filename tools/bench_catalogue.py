"""Catalogue redshifts -> comoving distances (fiducial.TabulatedDESI and DESI()), 10^7 redshifts resident in HBM: python tools/bench_catalogue.py"""
import os
import sys
import time
import warnings

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main(n=10**7, reps=10):
    import torch
    from cosmoprimo_amd.fiducial import DESI, TabulatedDESI
    from cosmoprimo_amd.utils import DistanceToRedshift
    warnings.simplefilter('ignore')
    dev = torch.device('cuda:0')
    z = torch.rand(n, dtype=torch.float64, device=dev) * 3.
    zf = z.to(torch.float32)
    tab, cosmo = TabulatedDESI(), DESI(engine='eisenstein_hu')
    d2z = DistanceToRedshift(cosmo.comoving_radial_distance)
    dist = cosmo.comoving_radial_distance(z)
    cases = [('TabulatedDESI().comoving_radial_distance(z f64 on the device)', lambda: tab.comoving_radial_distance(z), 16),
             ('TabulatedDESI().comoving_radial_distance(z f32 on the device)', lambda: tab.comoving_radial_distance(zf), 8),
             ('TabulatedDESI().efunc(z)', lambda: tab.efunc(z), 16),
             ('DESI().comoving_radial_distance(z)', lambda: cosmo.comoving_radial_distance(z), 16),
             ('DESI().luminosity_distance(z)', lambda: cosmo.luminosity_distance(z), 16),
             ('DistanceToRedshift(DESI().comoving_radial_distance)(distances)', lambda: d2z(dist), 16)]
    for name, fn, nbytes in cases:
        try:
            for _ in range(3):
                out = fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                out = fn()
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / reps * 1e3
            where = 'device tensor' if torch.is_tensor(out) else type(out).__name__
            print('%-70s %8.3f ms  %7.1f M/s  %6.0f GB/s algorithmic  -> %s %s' % (name, ms, n / ms / 1e3, n * nbytes / ms / 1e6, where, getattr(out, 'dtype', '')))
        except Exception as exc:
            print('%-70s FAILED %s: %s' % (name, type(exc).__name__, str(exc)[:100]))


if __name__ == '__main__':
    main()
