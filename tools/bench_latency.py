"""Latency of single-cosmology calls (BASELINE config 1's regime: one transform, one cosmology, what a sampler does per step), this package on the GPU
against the reference on the host's CPU with the same script.

    python tools/bench_latency.py                 # cosmoprimo_amd (needs the GPU)
    python tools/bench_latency.py --reference     # the reference imported from /root/reference (build container only), numpy on one core

Each line: median and minimum wall time of the call over ``--reps`` repetitions after two untimed ones, results brought to the host (what a caller gets).
"""
import argparse
import importlib
import sys
import os
import time
import warnings

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def timed(fn, reps):
    for _ in range(2):
        fn()
    times = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        times.append(time.perf_counter() - t0)
    return np.median(times) * 1e3, np.min(times) * 1e3


def main():
    parser = argparse.ArgumentParser()
    parser.add_argument('--reference', action='store_true')
    parser.add_argument('--reps', type=int, default=20)
    args = parser.parse_args()
    warnings.simplefilter('ignore')
    if args.reference:
        sys.path.insert(0, ROOT)
        from oracle._refimport import import_reference       # /root/reference, build container only
        cp = import_reference()
        from cosmoprimo.fftlog import PowerToCorrelation, TophatVariance
        from cosmoprimo.bao_filter import PowerSpectrumBAOFilter
        sync = lambda: None
    else:
        sys.path.insert(0, ROOT)
        cp = importlib.import_module('cosmoprimo_amd')
        from cosmoprimo_amd.fftlog import PowerToCorrelation, TophatVariance
        from cosmoprimo_amd.bao_filter import PowerSpectrumBAOFilter
        import torch
        sync = lambda: torch.cuda.synchronize()

    host = lambda x: np.asarray(x)
    k1024 = np.geomspace(1e-5, 1e2, 1024)
    cosmo = cp.Cosmology(engine='eisenstein_hu')
    fo = cosmo.get_fourier()
    pk2d = fo.pk_interpolator()
    pk1d = pk2d.to_1d(z=0.)
    pk_values = host(pk1d(k1024))
    kq, zq, rq = np.geomspace(1e-3, 1., 50), np.array([0., 0.5, 1.]), np.geomspace(1., 100., 256)
    z100 = np.linspace(0., 3., 100)
    fftlog = PowerToCorrelation(k1024)
    tophat = TophatVariance(k1024)
    ba = cosmo.get_background()
    state = {'i': 0}

    def fresh_cosmology_step():
        state['i'] += 1
        c = cp.Cosmology(engine='eisenstein_hu', Omega_m=0.3 + 1e-4 * (state['i'] % 50), sigma8=0.8)
        p = c.get_fourier().pk_interpolator()
        out = [host(p.sigma8_z(0.)), host(c.get_background().comoving_radial_distance(zq)), host(c.get_thermodynamics().rs_drag)]
        out.append(host(PowerSpectrumBAOFilter(p.to_1d(z=0.), engine='wallish2018').pknow))
        return out

    cases = [
        ('FFTlog: PowerToCorrelation(k1024)(pk), one row (plan built before)', lambda: host(fftlog(pk_values)[1])),
        ('FFTlog: PowerToCorrelation(k1024) built + one row', lambda: host(PowerToCorrelation(k1024)(pk_values)[1])),
        ('FFTlog: TophatVariance(k1024)(pk), one row', lambda: host(tophat(pk_values)[1])),
        ('pk_interpolator()(50 k, 3 z)', lambda: host(pk2d(kq, zq))),
        ('pk_interpolator().sigma8_z(0)', lambda: host(pk2d.sigma8_z(0.))),
        ('pk_interpolator().sigma_rz(256 r, 64 z)', lambda: host(pk2d.sigma_rz(rq, np.linspace(0., 3., 64)))),
        ('to_1d(z=0).sigma_r(256 r)', lambda: host(pk1d.sigma_r(rq))),
        ('pk_interpolator().to_xi() built + (50 s, 3 z)', lambda: host(pk2d.to_xi()(np.geomspace(1., 150., 50), zq))),
        ('background.comoving_radial_distance(100 z)', lambda: host(ba.comoving_radial_distance(z100))),
        ('PowerSpectrumBAOFilter(pk1d, wallish2018).pknow', lambda: host(PowerSpectrumBAOFilter(pk1d, engine='wallish2018').pknow)),
        ('PowerSpectrumBAOFilter(pk1d, peakaverage, cosmo, cosmo_fid).pknow', lambda: host(PowerSpectrumBAOFilter(pk1d, engine='peakaverage', cosmo=cosmo, cosmo_fid=cosmo).pknow)),
        ('PowerSpectrumBAOFilter(pk1d, brieden2022, cosmo, cosmo_fid).pknow', lambda: host(PowerSpectrumBAOFilter(pk1d, engine='brieden2022', cosmo=cosmo, cosmo_fid=cosmo).pknow)),
        ('PowerSpectrumBAOFilter(pk1d, savgol).pknow', lambda: host(PowerSpectrumBAOFilter(pk1d, engine='savgol').pknow)),
        ('sampler step: fresh Cosmology(Omega_m, sigma8) -> sigma8_z, distances, rs_drag, wallish2018 pknow', fresh_cosmology_step),
    ]
    print('%s, %d repetitions per line' % ('reference (numpy, CPU)' if args.reference else 'cosmoprimo_amd (MI355X)', args.reps))
    for name, fn in cases:
        try:
            def run():
                out = fn()
                sync()
                return out
            med, low = timed(run, args.reps)
            print('%-100s median %9.3f ms   min %9.3f ms' % (name, med, low))
        except Exception as exc:
            print('%-100s FAILED: %s: %s' % (name, type(exc).__name__, str(exc)[:80]))


if __name__ == '__main__':
    main()
