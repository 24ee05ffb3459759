"""Latency of small calls through the Python API (what an MCMC likelihood does once per step), this package on one MI355X.
python tools/latency_api.py [reference]      # 'reference': time /root/reference instead (build container only)"""
import os
import sys
import time
import warnings

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
warnings.simplefilter('ignore')


def main():
    use_ref = len(sys.argv) > 1 and sys.argv[1] == 'reference'
    if use_ref:
        from oracle._refimport import import_reference
        cp = import_reference()
        sync = lambda: None
    else:
        import torch
        import cosmoprimo_amd as cp
        sync = torch.cuda.synchronize
    k = np.geomspace(1e-3, 1., 200)
    z = np.linspace(0., 2., 10)

    def timeit(name, fn, n=50):
        fn()
        sync()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        sync()
        print('%-58s %9.1f us' % (name, (time.perf_counter() - t0) / n * 1e6))

    cosmo = cp.Cosmology(engine='eisenstein_hu')
    ba, fo = cosmo.get_background(), cosmo.get_fourier()
    pk = fo.pk_interpolator()
    timeit('Cosmology(engine=eisenstein_hu) + get_fourier (sigma8 norm)', lambda: cp.Cosmology(engine='eisenstein_hu', Omega_m=0.31).get_fourier(), 10)
    timeit('clone(Omega_m=...) + get_background', lambda: cosmo.clone(Omega_m=0.31).get_background(), 20)
    timeit('comoving_radial_distance(0.5)', lambda: ba.comoving_radial_distance(0.5))
    timeit('comoving_radial_distance(10 z)', lambda: ba.comoving_radial_distance(z))
    timeit('efunc(10 z)', lambda: ba.efunc(z))
    timeit('growth_factor(10 z)', lambda: ba.growth_factor(z))
    timeit('pk_interpolator()(200 k, 10 z)', lambda: pk(k, z))
    timeit('pk.sigma8_z(10 z)', lambda: pk.sigma8_z(z))
    timeit('pk.sigma_rz(8 r, 10 z)', lambda: pk.sigma_rz(np.linspace(2., 30., 8), z))
    timeit('fo.pk_interpolator() (new object)', lambda: fo.pk_interpolator())
    timeit('pk.to_1d(z=0.5)(200 k)', lambda: pk.to_1d(z=0.5)(k))
    timeit('PowerSpectrumBAOFilter(wallish2018) on one P(k)', lambda: cp.PowerSpectrumBAOFilter(pk.to_1d(z=0.), engine='wallish2018').pknow, 10)
    timeit('pk.to_1d(z=0).to_xi()(40 s)', lambda: pk.to_1d(z=0.).to_xi()(np.geomspace(1., 150., 40)), 10)
    timeit("cosmo['theta_cosmomc'] (fresh clone)", lambda: cosmo.clone(h=0.69)['theta_cosmomc'], 10)


if __name__ == '__main__':
    main()
