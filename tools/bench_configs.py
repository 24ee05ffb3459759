"""
Secondary measurements: BASELINE.json configs 3, 4 (wallish2018, one GPU's share) and 5 on ONE MI355X, inputs resident in HBM.
Prints one JSON line per config (not the driver's bench contract: that is bench.py, config 2).

    python tools/bench_configs.py [--scale 1.0]
"""
import argparse
import json
import os
import sys
import time
import warnings

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def timed(fn, reps, torch):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def cpu_config3(par, r, z, n):
    """The reference's path for one cosmology at a time (numpy port, oracle/): P(k, z) = P(k) growth^2(z) on 64 redshifts -> 64 FFTLogs
    (TophatVariance) -> natural spline to r -> sqrt; seconds per cosmology on one core."""
    from oracle import background as ob, power as op, sigma as osig
    t0 = time.perf_counter()
    for i in range(n):
        Om, Ob, h, ns = (float(par[name][i]) for name in ('Omega_m', 'Omega_b', 'h', 'n_s'))
        bg = ob.derived(h=h, Omega_b=Ob, Omega_m=Om)
        g2 = op.growth_factor(z, bg, znorm=0.)**2
        osig.sigma_r2(r, lambda k: op.pk_z0(k, 'eisenstein_hu', h=h, Omega_cdm=Om - Ob, Omega_b=Ob, n_s=ns)[:, None] * g2[None, :])**0.5
    return (time.perf_counter() - t0) / n


def cpu_config4(par, n):
    """wallish2018 of the numpy port for one P(k) vector at a time; seconds per vector on one core."""
    from oracle import bao as obao, power as op
    t0 = time.perf_counter()
    for i in range(n):
        Om, Ob, h, ns = (float(par[name][i]) for name in ('Omega_m', 'Omega_b', 'h', 'n_s'))
        obao.wallish2018(lambda k: op.pk_z0(k, 'eisenstein_hu', h=h, Omega_cdm=Om - Ob, Omega_b=Ob, n_s=ns)[:, None])
    return (time.perf_counter() - t0) / n


def cpu_config5(om, w0, wa, zz, n):
    """comoving_radial_distance of the numpy port, one fresh cosmology (119-knot table + natural spline) per sample; seconds per sample."""
    from oracle import background as ob
    t0 = time.perf_counter()
    for i in range(n):
        ob.comoving_radial_distance(np.array([zz[i]]), ob.derived(Omega_m=om[i], w0_fld=w0[i], wa_fld=wa[i]))
    return (time.perf_counter() - t0) / n


def baseline(seconds_per_unit, unit, sample):
    return {'value': 1. / seconds_per_unit, 'unit': unit, 'cores': 1, 'kind': 'port', 'sample': sample}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--scale', type=float, default=1.)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--chunk', type=int, default=16384, help='cosmologies per call in config 4')
    args = ap.parse_args()
    import torch
    import cosmoprimo_amd as cp
    from cosmoprimo_amd import background
    from cosmoprimo_amd.bao_filter import PowerSpectrumBAOFilter
    warnings.simplefilter('ignore')
    dev = torch.device('cuda:0')

    # config 3: sigma_rz on 256 r x 64 z, batch of EH cosmologies
    nb = int(10000 * args.scale)
    rng = np.random.default_rng(1)
    par = dict(Omega_m=rng.uniform(.25, .40, nb), Omega_b=rng.uniform(.04, .06, nb), h=rng.uniform(.6, .8, nb), n_s=rng.uniform(.92, 1., nb), sigma8=0.8)
    r, z = np.geomspace(1, 100, 256), np.linspace(0, 3, 64)
    t0 = time.perf_counter()
    cosmo = cp.Cosmology(engine='eisenstein_hu', **{k: (torch.as_tensor(v, device=dev) if np.ndim(v) else v) for k, v in par.items()})
    interp = cosmo.get_fourier().pk_interpolator()
    torch.cuda.synchronize()
    t_setup = time.perf_counter() - t0
    rt, zt = torch.as_tensor(r, device=dev), torch.as_tensor(z, device=dev)
    dt = timed(lambda: interp.sigma_rz(rt, zt), 3, torch)
    out_bytes = nb * 256 * 64 * 8
    print(json.dumps({'config': 3, 'workload': 'sigma_rz 256 r x 64 z, %d EH cosmologies (method fftlog, nk=1024; P(k, z) = P(k) x growth(z): one FFTLog per cosmology, growth applied to sigma^2)' % nb,
                      'value': nb / dt, 'unit': 'cosmologies/s', 'ms': dt * 1e3, 'setup_incl_sigma8_normalisation_ms': t_setup * 1e3,
                      'algorithmic_GBps': (out_bytes + nb * 80) / dt / 1e9,
                      'cpu_baseline': None if args.no_cpu_baseline else baseline(cpu_config3(par, r, z, 8), 'cosmologies/s', '8 cosmologies, one at a time')}))
    del interp, cosmo
    torch.cuda.empty_cache()

    # config 4 (one GPU's share of 1M vectors): wallish2018 on EH P(k) vectors, chunks of 16384 cosmologies
    nb = int(125000 * args.scale)
    rng = np.random.default_rng(2)
    par = dict(Omega_m=rng.uniform(.25, .40, nb), Omega_b=rng.uniform(.04, .06, nb), h=rng.uniform(.6, .8, nb), n_s=rng.uniform(.92, 1., nb))
    chunk = args.chunk

    filters = {}

    def one_chunk(sl, engine, **kw):
        """One chunk of cosmologies through filter ``engine``; the filter object is made once and called again for the next chunks, as a
        caller of the reference would do (``filter(pk_interpolator, cosmo=cosmo)``): its ``_prepare`` products depend on the fiducial only."""
        cosmo = cp.Cosmology(engine='eisenstein_hu', sigma8=0.8, **{k: torch.as_tensor(v[sl], device=dev) for k, v in par.items()})
        interp = cosmo.get_fourier().pk_interpolator(z=np.array([0.]))
        if engine not in filters:
            filters[engine] = PowerSpectrumBAOFilter(interp, engine=engine, **(dict(kw, cosmo=cosmo) if kw else {}))
        else:
            filters[engine](interp, cosmo=cosmo if kw else None)
        return filters[engine].pknow.shape[0]

    one_chunk(slice(0, min(nb, chunk)), 'wallish2018')    # untimed: plans, operators and library kernels are built on first use
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    done = 0
    for start in range(0, nb, chunk):
        done += one_chunk(slice(start, min(nb, start + chunk)), 'wallish2018')
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(json.dumps({'config': 4, 'filter': 'wallish2018',
                      'workload': 'wallish2018 on %d EH98 P(k) vectors (nk=1024), incl. P(k) generation + sigma8 normalisation + D2H of pknow' % nb,
                      'value': done / dt, 'unit': 'vectors/s', 'ms': dt * 1e3, 'algorithmic_GBps': done * 16384 / dt / 1e9,
                      'cpu_baseline': None if args.no_cpu_baseline else baseline(cpu_config4(par, 32), 'vectors/s', '32 vectors, one at a time, P(k) generation included')}))
    fid = cp.Cosmology(engine='eisenstein_hu')
    one_chunk(slice(0, min(nb, chunk)), 'brieden2022', cosmo_fid=fid)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    done = 0
    for start in range(0, nb, chunk):
        done += one_chunk(slice(start, min(nb, start + chunk)), 'brieden2022', cosmo_fid=fid)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(json.dumps({'config': 4, 'filter': 'brieden2022',
                      'workload': 'brieden2022 on %d EH98 P(k) vectors (one rs_drag ratio each), incl. P(k) + no-wiggle template generation, both sigma8 normalisations, D2H' % nb,
                      'value': done / dt, 'unit': 'vectors/s', 'ms': dt * 1e3, 'algorithmic_GBps': done * 16384 / dt / 1e9}))
    filters.clear()
    torch.cuda.empty_cache()

    # config 5 (one GPU's share of 10M samples): comoving_radial_distance for (Omega_m, w0, wa, z) samples
    nb = int(1250000 * args.scale)
    rng = np.random.default_rng(3)
    host5 = (rng.uniform(0.1, 0.5, nb), rng.uniform(-1.5, -0.5, nb), rng.uniform(-1., 0.5, nb), rng.uniform(0., 3., nb))
    om, w0, wa, zz = (torch.as_tensor(v, device=dev) for v in host5)
    dt = timed(lambda: background.distance('comoving_radial_distance', zz[:, None], dict(w0_fld=w0, wa_fld=wa), Omega_m=om, per_cosmology_z=True), 5, torch)
    print(json.dumps({'config': 5, 'workload': 'comoving_radial_distance, %d (Omega_m, w0, wa, z) samples, one fresh cosmology per sample' % nb,
                      'value': nb / dt, 'unit': 'samples/s', 'ms': dt * 1e3, 'algorithmic_GBps': nb * 40 / dt / 1e9,
                      'E_evaluations_per_s': nb * 237 / dt,
                      'cpu_baseline': None if args.no_cpu_baseline else baseline(cpu_config5(*host5, 2000), 'samples/s', '2000 samples, one cosmology each')}))

    # tabulated engine: redshift -> comoving distance for a catalogue (fiducial.TabulatedDESI, linear interpolation in a 40 002-row table)
    from cosmoprimo_amd.fiducial import TabulatedDESI
    tab = TabulatedDESI()
    nb = int(2e8 * args.scale)
    zcat = torch.rand(nb, device=dev, dtype=torch.float64) * 3.
    dt = timed(lambda: tab.comoving_radial_distance(zcat), 5, torch)
    line = {'config': 'tabulated', 'workload': 'TabulatedDESI().comoving_radial_distance, %d redshifts resident in HBM (range check included)' % nb,
            'value': nb / dt, 'unit': 'redshifts/s', 'ms': dt * 1e3, 'algorithmic_GBps': nb * 16 / dt / 1e9}
    if not args.no_cpu_baseline:
        table = tab.engine
        zh = np.random.default_rng(5).uniform(0., 3., 2000000)
        t0 = time.perf_counter()
        np.interp(zh, table.z, table.comoving_radial_distance)
        line['cpu_baseline'] = baseline((time.perf_counter() - t0) / zh.size, 'redshifts/s', 'numpy.interp, 2e6 redshifts (the reference\'s own call)')
        line['cpu_baseline']['kind'] = 'reference'
    print(json.dumps(line))
    # and back: distance -> redshift for the same catalogue (utils.DistanceToRedshift: a cubic spline of z(D_C) at every distance)
    from cosmoprimo_amd.utils import DistanceToRedshift
    dcat = tab.comoving_radial_distance(zcat)
    redshift = DistanceToRedshift(distance=tab.comoving_radial_distance, zmax=10., nz=4096)
    dt = timed(lambda: redshift(dcat), 5, torch)
    err = float((redshift(dcat) - zcat).abs().max())
    print(json.dumps({'config': 'distance_to_redshift', 'workload': 'DistanceToRedshift (4096-knot natural spline) at %d resident distances' % nb,
                      'value': nb / dt, 'unit': 'distances/s', 'ms': dt * 1e3, 'algorithmic_GBps': nb * 16 / dt / 1e9, 'max_abs_redshift_error': err}))
    del zcat, dcat
    torch.cuda.empty_cache()

    # f4: the batch driver (emulators.get_calculator): params -> every section's arrays on the reference's default grids, D2H included
    from cosmoprimo_amd.emulators import get_calculator
    nb = int(8192 * args.scale)
    rng = np.random.default_rng(4)
    par = dict(Omega_m=rng.uniform(.25, .40, nb), Omega_b=rng.uniform(.04, .06, nb), h=rng.uniform(.6, .8, nb), n_s=rng.uniform(.92, 1., nb))
    calc = get_calculator(cp.Cosmology(engine='eisenstein_hu'))
    calc(**{k: v[:64] for k, v in par.items()})
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = calc(**par)
    dt = time.perf_counter() - t0
    nbytes = sum(v.nbytes for v in out.values())
    print(json.dumps({'config': 'f4', 'workload': 'get_calculator(eisenstein_hu)(%d cosmologies): background (256 z) + thermodynamics + primordial + 3 P(k, z) pairs (422 k x 30 z), results on the host' % nb,
                      'value': nb / dt, 'unit': 'cosmologies/s', 'ms': dt * 1e3, 'output_GB': nbytes / 1e9,
                      'reference_cpu_note': 'the reference calculator takes 0.67 s per cosmology in the build container (without the P(k) pairs, which it drops)'}))


if __name__ == '__main__':
    main()
