"""Generate golden vectors from the imported reference (run in the build container only).

    python -m oracle.gen_golden [target ...]      # default target: fftlog

Targets -> tests/golden/<name>.npz: fftlog (tables, loggamma, transforms), background, power, sigma, sigma_quad, sigma_api (the module-level sigma integrals with the reference's arguments), api_signatures (json: signatures of the public surface), bao, xi, bao2, bspline, densities, ncdm, variants, power_ncdm, bao_batch, fuzz (random cosmologies from wide priors), fftlog_fuzz (random FFTLog configurations), fftlog_large (the same at padded lengths 16 384 ... 131 072), interp_fuzz (tabulated interpolators with random options), filter_fuzz (BAO filters with random options), params_fuzz (parameter conventions), xi_fuzz (tabulated xi interpolators with random options),
calculator, cosmology_api, api_flows (tests/api_scenarios.py replayed with the reference), abacus (also writes the package data cosmoprimo_amd/data/abacus_cosmologies.json), desi_table (161 rows of the
reference's data/desi.dat).  Every vector is the output of the reference itself, imported from /root/reference; no reference source is stored.
See SURVEY.md 8(c) for the list (G1..G8).  TEST INFRASTRUCTURE: the product never imports this module.
"""
import os
import sys
import warnings
import numpy as np

from ._refimport import import_reference, REFERENCE_ROOT

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests', 'golden')


def save(name, **arrays):
    path = os.path.join(OUT, name + '.npz')
    np.savez_compressed(path, **arrays)
    print('wrote %s (%.1f KB)' % (path, os.path.getsize(path) / 1e3))


def gen_pk_eh(cp):
    cosmo = cp.Cosmology()
    fo = cp.Fourier(cosmo, engine='eisenstein_hu')
    interp = fo.pk_interpolator()
    out = {}
    for n in (1024, 2048):
        k = np.logspace(-5, 2, n)
        out['k%d' % n] = k
        out['pk%d' % n] = interp(k, z=0)
    save('pk_eh_default', **out)
    return out


def _tables(f, every=1):
    # 'every' > 1: strided spot samples (keeps the fixture set small); y/u strided by every//4
    ev4 = max(every // 4, 1)
    return dict(delta=np.asarray(f.delta), lnxy=np.asarray(f.lnxy), y=np.asarray(f.y)[..., ::ev4], u=np.asarray(f.padded_u)[..., ::ev4],
                pre=np.asarray(f.padded_prefactor)[..., ::every], post=np.asarray(f.padded_postfactor)[..., ::every],
                sizes=np.array([f.padded_size, f.padded_size_in_left, f.padded_size_in_right, f.padded_size_out_left, f.padded_size_out_right]))


def gen_fftlog_tables(cp):
    """G1: tables of every FFTlog subclass at N in {1024, 2048} (+ the test_pad case)."""
    from cosmoprimo import fftlog as fl
    out = {}
    for n in (1024, 2048):
        k = np.logspace(-5, 2, n)
        cases = {
            'p2c_l0': fl.PowerToCorrelation(k, ell=0),
            'p2c_multi': fl.PowerToCorrelation(k, ell=[0, 1, 2, 3, 4]),
            'p2c_multi_cplx': fl.PowerToCorrelation(k, ell=[0, 1, 2, 3, 4], complex=True),
            'c2p_l0': fl.CorrelationToPower(k, ell=0),
            'c2p_l2_q': fl.CorrelationToPower(k, ell=2, q=0.5),
            'tophat': fl.TophatVariance(k),
            'gauss': fl.GaussianVariance(k),
            'hankel_nu0_q1': fl.HankelTransform(k, nu=0, q=1),
            'hankel_nu2': fl.HankelTransform(k, nu=[0, 2], q=1),
            'p2c_nolowring': fl.PowerToCorrelation(k, ell=0, lowring=False, xy=1.),
        }
        for name, f in cases.items():
            every = 1 if name in ('p2c_l0', 'tophat') else 16
            for key, val in _tables(f, every).items():
                out['n%d_%s_%s' % (n, name, key)] = val
    x = np.logspace(-3, 3, num=7, endpoint=True)   # reference tests/test_fftlog.py:40-45
    f = fl.HankelTransform(x, minfolds=3, xy=1, lowring=False)
    for key, val in _tables(f).items():
        out['pad7_%s' % key] = val
    out['pad7_padded_x'], out['pad7_padded_y'] = f.padded_x, f.padded_y
    # raw FFTlog with generic kernels (TophatKernel, TophatSqKernel ndim=1,2, GaussianKernel, BesselJKernel)
    x = np.logspace(-4, 3, 200)
    gen = {
        'tophat1': fl.TophatKernel(ndim=1), 'tophat3': fl.TophatKernel(ndim=3), 'tophatsq1': fl.TophatSqKernel(ndim=1),
        'tophatsq2': fl.TophatSqKernel(ndim=2), 'gaussian': fl.GaussianKernel(), 'besselj1': fl.BesselJKernel(1.5),
        'sphbesselj3': fl.SphericalBesselJKernel(3),
    }
    for name, kern in gen.items():
        f = fl.FFTlog(x, kern, q=0.7, minfolds=3)
        for key, val in _tables(f).items():
            out['gen200_%s_%s' % (name, key)] = val
    save('fftlog_tables', **out)


def gen_loggamma():
    """G2: scipy loggamma/gamma on the arguments the kernels use + a stress grid."""
    from scipy.special import loggamma, gamma
    rng = np.random.default_rng(42)
    re = np.concatenate([rng.uniform(-2.5, 3., 1500), np.linspace(-2.45, 2.95, 100)])
    im = np.concatenate([rng.uniform(0., 450., 1000), rng.uniform(0., 3., 500), np.zeros(50), rng.uniform(-450, 0., 50)])
    z = re + 1j * im
    # exact arguments of the N=2048 p2c ell=0 u-table: 0.5*(nu + q + i t) and 0.5*(3 + nu - q - i t)
    d = np.log(1e7) / 2047
    t = 2 * np.pi / 4096 / d * np.arange(2049)
    zz = np.concatenate([0.5 * (1.5 + 1j * t), 0.5 * (1.5 - 1j * t), 0.5 * (1.5 - 4 + 1j * t), 0.5 * (5 - 1.5 - 1j * t)])
    z = np.concatenate([z, zz])
    zg = np.concatenate([rng.uniform(0.05, 3., 200) + 1j * rng.uniform(-40., 40., 200), 0.5 * (1.5 + 1j * t[::8])])
    save('loggamma', z=z, loggamma=loggamma(z), zg=zg, gamma=gamma(zg))


def gen_fftlog_transforms(cp, pks):
    """G3: transforms of the default EH P(k) (configs 1 and 2) + config-2 batch rows + analytic Hankel pair."""
    from cosmoprimo import fftlog as fl
    from .workloads import config2_rows
    out = {}
    for n in (1024, 2048):
        k, pk = pks['k%d' % n], pks['pk%d' % n]
        f = fl.PowerToCorrelation(k, ell=0, lowring=True)
        for name, extrap in [('zero', 0), ('edge', 'edge'), ('log', 'log'), ('mixed', ('log', 0.)), ('const', (1.5, 'edge'))]:
            s, xi = f(pk, extrap=extrap)
            out['n%d_p2c_l0_%s' % (n, name)] = xi
        out['n%d_p2c_l0_s' % n] = s
        out['n%d_p2c_l0_keep' % n] = f(pk, extrap='log', keep_padding=True)[1]
        fm = fl.PowerToCorrelation(k, ell=[0, 2, 4], lowring=True)
        out['n%d_p2c_l024' % n] = fm(pk)[1]
        out['n%d_p2c_l024_s' % n] = fm(pk)[0]
        ft = fl.TophatVariance(k)
        out['n%d_tophat' % n] = ft(pk)[1]
        out['n%d_tophat_r' % n] = ft(pk)[0]
        fc = fl.CorrelationToPower(s, ell=0, lowring=True)
        out['n%d_c2p_l0' % n] = fc(out['n%d_p2c_l0_zero' % n])[1]
        out['n%d_c2p_l0_k' % n] = fc(out['n%d_p2c_l0_zero' % n])[0]
    # config-2 batch rows 0, 1, 49999, 99999
    k, pk = pks['k2048'], pks['pk2048']
    f = fl.PowerToCorrelation(k, ell=0, lowring=True)
    idx = np.array([0, 1, 49999, 99999])
    rows = np.concatenate([config2_rows(k, pk, i, i + 1) for i in idx])
    out['config2_idx'] = idx
    out['config2_xi'] = f(rows)[1]
    # analytic Hankel pair (reference tests/test_fftlog.py:58-81), incl. inv() and batched input
    x = np.logspace(-3, 3, num=60, endpoint=False)
    fx = 1 / (1 + x**2)**1.5
    hf = fl.HankelTransform(x, nu=0, q=1, lowring=True)
    y, g = hf(fx, extrap='log')
    out['hankel60_x'], out['hankel60_f'], out['hankel60_y'], out['hankel60_g'] = x, fx, y, g
    hf.inv()
    x2, f2 = hf(g, extrap='log')
    out['hankel60_inv_x'], out['hankel60_inv_f'] = x2, f2
    for key, val in _tables(hf).items():
        out['hankel60_inv_%s' % key] = val
    save('fftlog_transforms', **out)


def gen_background(cp):
    """G8: 32 cosmologies (Omega_m, w0, wa, Omega_k, h): the 119-knot D_C table, E(z) and D_C/D_M/D_A/D_L at 64 z."""
    import warnings
    rng = np.random.default_rng(8)
    n = 32
    par = dict(Omega_m=rng.uniform(0.1, 0.5, n), w0_fld=rng.uniform(-1.5, -0.5, n), wa_fld=rng.uniform(-1., 0.5, n),
               Omega_k=np.where(np.arange(n) % 3 == 0, 0., rng.uniform(-0.1, 0.1, n)), h=rng.uniform(0.6, 0.8, n), Omega_b=np.full(n, 0.05))
    par['w0_fld'][0], par['wa_fld'][0] = -1., 0.    # one plain LCDM
    z = np.concatenate([[0., 1e-4, 5e-3], np.linspace(0.01, 3., 56), [10., 100., 1100., 5000., 9999.]])
    out = {k: v for k, v in par.items()}
    out['z'] = z
    tabs, es, ds = [], [], {name: [] for name in ['comoving_radial_distance', 'comoving_transverse_distance', 'angular_diameter_distance', 'luminosity_distance']}
    for i in range(n):
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            cosmo = cp.Cosmology(engine='bbks', **{k: float(v[i]) for k, v in par.items()})
            ba = cosmo.get_background()
            for name in ds:
                ds[name].append(getattr(ba, name)(z))
            es.append(ba.efunc(z))
            tabs.append(np.asarray(ba._cache['comoving_radial_distance']._fun))
            if i == 0:
                out['zc'] = np.asarray(ba._cache['comoving_radial_distance']._x)
                out['nan_outside'] = ba.comoving_radial_distance(np.array([-0.1, 1e4]))
                out['f4'] = ba.comoving_radial_distance(np.linspace(0., 2., 5).astype('f4'))
    out['table'] = np.array(tabs)
    out['efunc'] = np.array(es)
    for name, v in ds.items():
        out[name] = np.array(v)
    save('background', **out)


POWER_PARAMS = dict(Omega_m=[0.31, 0.25, 0.40, 0.28, 0.35, 0.30, 0.27, 0.33], Omega_b=[0.045, 0.04, 0.06, 0.05, 0.055, 0.048, 0.042, 0.05],
                    h=[0.68, 0.6, 0.8, 0.7, 0.72, 0.65, 0.75, 0.67], n_s=[0.97, 0.92, 1.0, 0.96, 0.95, 0.98, 0.93, 0.965],
                    sigma8=[0.81, 0.8, 0.8, 0.75, 0.9, 0.8, 0.85, 0.8], alpha_s=[0.01, 0., 0., -0.02, 0., 0., 0., 0.],
                    w0_fld=[-1., -1., -0.9, -1., -1.1, -1., -1., -0.8], wa_fld=[0., 0., 0.2, 0., -0.3, 0., 0., 0.1])


def gen_power(cp):
    """G7: 8 cosmologies x 3 analytic engines: scalars, transfer_k, pk_k, P(k, z), growth factor / rate, sigma8 rescale factor."""
    import warnings
    k = np.concatenate([np.logspace(-7, 2, 46), [1e-5, 0.05, 0.1, 1.]])
    z = np.array([0., 0.5, 1., 2., 3.])
    out = {name: np.array(v, dtype='f8') for name, v in POWER_PARAMS.items()}
    out['k'], out['z'] = k, z
    n = len(POWER_PARAMS['h'])
    names_sc = ['z_eq', 'k_eq', 'z_drag', 'r_drag', 'r_eq', 'rs_drag', 'k_silk', 'alpha_c', 'beta_c', 'alpha_b', 'beta_node', 'beta_b']
    for eng in ['eisenstein_hu', 'eisenstein_hu_nowiggle', 'bbks']:
        acc = {key: [] for key in ['transfer', 'pk_prim', 'pkz', 'growth_factor', 'growth_factor_znorm0', 'growth_rate', 'rsigma8', 'A_s', 'sigma8_m']}
        sc = {key: [] for key in names_sc + ['alpha_gamma', 'gamma', 'rs_drag_h', 'z_drag_th']}
        for i in range(n):
            with warnings.catch_warnings():
                warnings.simplefilter('ignore')
                cosmo = cp.Cosmology(engine=eng, **{name: float(v[i]) for name, v in POWER_PARAMS.items()})
                fo, tr, pm, ba = cosmo.get_fourier(), cosmo.get_transfer(), cosmo.get_primordial(), cosmo.get_background()
                acc['transfer'].append(tr.transfer_k(k))
                acc['pk_prim'].append(pm.pk_k(k))
                acc['pkz'].append(fo.pk_interpolator()(k, z))
                acc['growth_factor'].append(ba.growth_factor(z))
                acc['growth_factor_znorm0'].append(ba.growth_factor(z, znorm=0.))
                acc['growth_rate'].append(ba.growth_rate(z))
                acc['rsigma8'].append(cosmo._engine._rsigma8)
                acc['A_s'].append(pm.A_s)
                acc['sigma8_m'].append(fo.sigma8_m)
                eng_obj = cosmo._engine
                for key in sc:
                    if key == 'rs_drag_h':
                        sc[key].append(cosmo.get_thermodynamics().rs_drag if eng != 'bbks' else np.nan)
                    elif key == 'z_drag_th':
                        sc[key].append(cosmo.get_thermodynamics().z_drag if eng != 'bbks' else np.nan)
                    else:
                        sc[key].append(getattr(eng_obj, key, np.nan))
        for key, v in {**acc, **sc}.items():
            out['%s_%s' % (eng, key)] = np.array(v, dtype='f8')
    save('power', **out)


def gen_sigma(cp):
    """G4: sigma_rz / sigma_dz / sigma8 of the EH callable interpolator and of a 500 x 30 table; fftlog and simpson methods; to_xi."""
    import warnings
    out = {}
    r = np.geomspace(1., 100., 256)
    z = np.linspace(0., 3., 64)
    out['r'], out['z'] = r, z
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        cosmo = cp.Cosmology(engine='eisenstein_hu')
        fo = cosmo.get_fourier()
        interp = fo.pk_interpolator()
        out['eh_sigma_rz'] = interp.sigma_rz(r, z)
        out['eh_sigma_rz_simpson'] = interp.sigma_rz(r[::16], z[::8], method='simpson')
        out['eh_sigma_dz'] = interp.sigma_dz(z)
        out['eh_sigma8_z'] = interp.sigma8_z(z)
        i1 = interp.to_1d(z=0.)
        out['eh_sigma_r_1d'] = i1.sigma_r(r)
        out['eh_sigma_d_1d'] = i1.sigma_d()
        xi = interp.to_xi()
        out['eh_xi_s'], out['eh_xi'] = np.asarray(xi.s), np.asarray(xi.xi)[..., ::8]
        xi1 = i1.to_xi()
        out['eh_xi1_s'], out['eh_xi1'] = np.asarray(xi1.s), np.asarray(xi1.xi)
        # table input (config 3 B): k = logspace(-4, 2, 500), z = linspace(0, 3, 30)
        kt, zt = np.logspace(-4, 2, 500), np.linspace(0., 3., 30)
        pkt = interp(kt, zt)
        out['table_k'], out['table_z'], out['table_pk'] = kt, zt, pkt
        tab = cp.PowerSpectrumInterpolator2D(kt, zt, pkt)
        out['table_sigma_rz'] = tab.sigma_rz(r, z)
        ke = np.geomspace(1e-7, 1e2, 64)
        out['table_eval_k'] = ke
        out['table_eval'] = tab(ke, z[::4])
        tab1 = cp.PowerSpectrumInterpolator1D(kt, pkt[:, 0])
        out['table1d_eval'] = tab1(ke)
        out['table1d_sigma_r'] = tab1.sigma_r(r[::8])
        out['table1d_sigma8'] = tab1.sigma8()
    save('sigma', **out)


def gen_sigma_quad(cp):
    """The sigma integrals with method='quad' (scipy.integrate.quad per (r, column), interpolator.py:167-177, 255-273) on the G4 table."""
    import warnings
    out = {}
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        interp = cp.Cosmology(engine='eisenstein_hu').get_fourier().pk_interpolator()
        kt, zt = np.logspace(-4, 2, 500), np.linspace(0., 3., 30)
        pkt = interp(kt, zt)
        out['table_k'], out['table_z'], out['table_pk'] = kt, zt, pkt
        out['r'], out['z'] = np.array([2., 8., 30.]), np.array([0., 1.45])
        tab1 = cp.PowerSpectrumInterpolator1D(kt, pkt[:, 0])
        out['sigma_r'] = tab1.sigma_r(out['r'], method='quad')
        out['sigma_d'] = tab1.sigma_d(method='quad')
        # the 2-D classes fail with method='quad' in the reference (interpolator.py:259 indexes p[:, i] on the 1-D pk(scalar k)):
        # one 1-D interpolator per redshift stands for them
        out['sigma_rz'] = np.stack([cp.PowerSpectrumInterpolator1D(kt, pkt[:, iz]).sigma_r(out['r'], method='quad') for iz in (0, 14)], axis=-1)
        out['sigma_dz'] = np.array([cp.PowerSpectrumInterpolator1D(kt, pkt[:, iz]).sigma_d(method='quad') for iz in (0, 14)])
        out['z'] = zt[[0, 14]]
        out['sigma_r_tight'] = tab1.sigma_r(out['r'], method='quad', epsabs=1e-10, epsrel=1e-10)
        # ... which stops at quad's default of 50 subintervals, 5e-6 short of the request; the same integrand with room to converge:
        from scipy import integrate
        from cosmoprimo.interpolator import kernel_tophat2
        limits = (np.log(1e-7 * (1. + 1e-9)), np.log(1e2 * (1. - 1e-9)))
        out['sigma_r_converged'] = np.array([(integrate.quad(lambda logk: kernel_tophat2(np.exp(logk) * rr) * np.exp(logk)**3 * tab1(np.exp(logk)), *limits,
                                                             epsabs=1e-12, epsrel=1e-12, limit=4000)[0] / (2. * np.pi**2))**0.5 for rr in out['r']])
    save('sigma_quad', **out)


def sigma_api_spectrum(k, ncol=0):
    """A smooth closed-form P(k) for the module-level sigma integrals (callable convention of the reference: (nk,) -> (nk,) or (nk, ncol));
    shared by the generator and tests/test_sigma_api_gpu.py."""
    k = np.asarray(k, dtype='f8')
    q = k / 0.02
    p = 2e4 * q**0.96 / (1. + 0.7 * q + (0.9 * q)**2)**1.9
    if ncol:
        return p[..., None] * (1. + 0.25 * np.arange(ncol))
    return p


def sigma_api_gaussian2(x):
    """W^2 of a Gaussian window, a ``kernel=`` other than the default."""
    return np.exp(-np.asarray(x, dtype='f8')**2)


def gen_sigma_api(cp):
    """integrate_sigma_r2 / integrate_sigma_d2 called as module-level functions with the reference's own arguments (interpolator.py:123, 200):
    callable P(k), scalar / nd radii, one or several columns, every method, a custom kernel, float32 radii."""
    from cosmoprimo.interpolator import integrate_sigma_r2, integrate_sigma_d2
    out = {}
    r1 = np.array([[2., 8., 30.], [4., 12., 50.]])
    for ncol in (0, 3):
        pk = lambda k: sigma_api_spectrum(k, ncol)      # noqa: E731
        # (method='quad' with several columns fails in the reference -- interpolator.py:259 indexes p[:, i] on the 1-D pk(scalar k): one column only)
        for method in ('fftlog', 'simpson', 'leggauss', 'quad')[:4 if ncol == 0 else 3]:
            out['r2_%s_%d' % (method, ncol)] = integrate_sigma_r2(r1, pk, method=method)
            out['r2s_%s_%d' % (method, ncol)] = integrate_sigma_r2(8., pk, method=method)
        for method in ('simpson', 'leggauss', 'quad')[:3 if ncol == 0 else 2]:
            out['d2_%s_%d' % (method, ncol)] = integrate_sigma_d2(pk, method=method)
            out['r2g_%s_%d' % (method, ncol)] = integrate_sigma_r2(r1, pk, method=method, kernel=sigma_api_gaussian2)
        out['r2_range_%d' % ncol] = integrate_sigma_r2(r1, pk, kmin=1e-5, kmax=10., nk=512, method='simpson')
        out['r2_f4_%d' % ncol] = integrate_sigma_r2(r1.astype('f4'), pk)
        # (its dtype is part of the fixture: float32 for one column -- pk(kmin) is a scalar and does not count, interpolator.py:247 --, float64 for several)
        out['d2_range_%d' % ncol] = integrate_sigma_d2(pk, kmin=1e-5, kmax=10., nk=512)
    out['r'] = r1
    save('sigma_api', **out)


API_MODULES = ['fftlog', 'interpolator', 'bao_filter', 'cosmology', 'eisenstein_hu', 'eisenstein_hu_nowiggle', 'eisenstein_hu_nowiggle_variants', 'bbks', 'utils']


def api_signature(obj):
    """Parameters of a callable as [name, kind, repr of the default or None]: data about a signature, no source text."""
    import inspect
    try:
        signature = inspect.signature(obj)
    except (TypeError, ValueError):
        return None
    out = []
    for par in signature.parameters.values():
        if par.default is inspect.Parameter.empty:
            default = None
        elif callable(par.default):
            default = 'callable:' + getattr(par.default, '__name__', '?')
        else:
            default = repr(par.default)
        out.append([par.name, par.kind.name, default])
    return out


def api_surface(package):
    """Public surface of the section-8(a) modules of ``package`` ('cosmoprimo' here, 'cosmoprimo_amd' in tests/test_api_signatures.py): per module the
    public functions (signature) and classes (constructor signature, and per public member -- inherited ones included -- its kind and signature),
    plus ``__all__`` of the package."""
    import importlib
    import inspect
    out = {}
    for name in API_MODULES:
        module = importlib.import_module(package + '.' + name)
        entries = {}
        for oname, obj in vars(module).items():
            if oname.startswith('_') or getattr(obj, '__module__', None) != module.__name__:
                continue
            if inspect.isclass(obj):
                members = {}
                for mname in dir(obj):
                    if mname.startswith('_') and mname not in ('__call__', '__init__'):
                        continue
                    try:
                        member = inspect.getattr_static(obj, mname)
                    except AttributeError:
                        continue
                    if isinstance(member, property):
                        members[mname] = {'kind': 'property'}
                        continue
                    kind, raw = 'method', member
                    if isinstance(member, staticmethod):
                        kind, raw = 'staticmethod', member.__func__
                    elif isinstance(member, classmethod):
                        kind, raw = 'classmethod', member.__func__
                    if inspect.isfunction(raw):
                        members[mname] = {'kind': kind, 'sig': api_signature(raw)}
                    elif not callable(raw) and not mname.startswith('_'):
                        members[mname] = {'kind': 'attribute'}
                entries[oname] = {'kind': 'class', 'init': api_signature(obj), 'members': members}
            elif inspect.isfunction(obj):
                entries[oname] = {'kind': 'function', 'sig': api_signature(obj)}
        out[name] = entries
    out['__init__'] = {'all': sorted(getattr(importlib.import_module(package), '__all__', []))}
    return out


def gen_api_signatures(cp):
    """tests/golden/api_signatures.json: ``inspect.signature`` of every public class / function / method of the section-8(a) modules of the reference."""
    import json
    path = os.path.join(OUT, 'api_signatures.json')
    with open(path, 'w') as f:
        json.dump(api_surface('cosmoprimo'), f, indent=0, sort_keys=True)
    print('wrote %s (%.1f KB)' % (path, os.path.getsize(path) / 1e3))


BAO_PARAMS = [dict(), dict(Omega_m=0.27, Omega_b=0.045, h=0.72, n_s=0.95), dict(Omega_m=0.36, Omega_b=0.055, h=0.64, n_s=0.98, sigma8=0.85),
              dict(Omega_m=0.31, Omega_b=0.049, h=0.6766, n_s=0.9665, sigma8=0.81)]


def gen_bao(cp):
    """G6: wallish2018 and brieden2022 on the EH98 P(k) of 4 cosmologies (1D callable input), a 4-column tabulated 2D input,
    wallish intermediates and brieden _prepare products."""
    import warnings
    out = {}
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        fid = cp.Cosmology(engine='eisenstein_hu')
        for i, par in enumerate(BAO_PARAMS):
            cosmo = cp.Cosmology(engine='eisenstein_hu', **par)
            interp = cosmo.get_fourier().pk_interpolator().to_1d(z=0.)
            w = cp.PowerSpectrumBAOFilter(interp, engine='wallish2018', cosmo=cosmo, cosmo_fid=fid)
            out['c%d_wallish_pknow' % i] = w.pknow
            out['c%d_pk' % i] = w.pk
            if i == 0:
                out['k'] = w.k
                for name in ['_even', '_odd', '_dd_even', '_dd_odd', '_even_now', '_odd_now']:
                    out['c0_wallish' + name] = getattr(w, name)
            b = cp.PowerSpectrumBAOFilter(interp, engine='brieden2022', cosmo=cosmo, cosmo_fid=fid)
            out['c%d_brieden_pknow' % i] = b.pknow
            out['c%d_rs_drag_ratio' % i] = b.rs_drag_ratio()
            if i == 0:
                out['brieden_k_fid'] = b.k_fid
                out['brieden_pknow_correction'] = b.pknow_correction
                out['brieden_ratio_fid'] = b.ratio_fid
                out['brieden_ratio_now_fid'] = b.ratio_now_fid
                out['brieden_peaks_high'], out['brieden_peaks_low'] = [np.asarray(ix) for ix in b.ik_fid_peaks]
            # the same with the 2D interpolator built on a single redshift (no growth factor in the filter input, bao_filter.py:96-98):
            # the input shape of batched runs
            interp2d = cosmo.get_fourier().pk_interpolator(z=np.array([0.]))
            out['c%d_wallish_pknow_2d' % i] = cp.PowerSpectrumBAOFilter(interp2d, engine='wallish2018', cosmo=cosmo, cosmo_fid=fid).pknow
            out['c%d_brieden_pknow_2d' % i] = cp.PowerSpectrumBAOFilter(interp2d, engine='brieden2022', cosmo=cosmo, cosmo_fid=fid).pknow
            # no cosmo given: rs_drag_ratio = 1
            if i == 1:
                out['c1_brieden_pknow_nocosmo'] = cp.PowerSpectrumBAOFilter(interp, engine='brieden2022', cosmo_fid=fid).pknow
                out['c1_wallish_pknow_nocosmo'] = cp.PowerSpectrumBAOFilter(interp, engine='wallish2018').pknow
        # tabulated 2D input with 4 redshifts (columns differ)
        cosmo = cp.Cosmology(engine='eisenstein_hu', **BAO_PARAMS[3])
        interp2 = cosmo.get_fourier().pk_interpolator()
        kt, zt = np.logspace(-5, 1.5, 400), np.array([0., 0.5, 1., 1.5])
        tab = cp.PowerSpectrumInterpolator2D(kt, zt, interp2(kt, zt))
        out['tab_k'], out['tab_z'], out['tab_pk'] = kt, zt, interp2(kt, zt)
        out['tab_wallish_pknow'] = cp.PowerSpectrumBAOFilter(tab, engine='wallish2018', cosmo=cosmo, cosmo_fid=fid).pknow
        out['tab_brieden_pknow'] = cp.PowerSpectrumBAOFilter(tab, engine='brieden2022', cosmo=cosmo, cosmo_fid=fid).pknow
    save('bao', **out)


def gen_xi(cp):
    """f1 (xi side): CorrelationFunctionInterpolator1D/2D built by to_xi() of the EH interpolators and from tables; evaluation,
    to_pk() round trip, sigma8 through to_pk; the 1D 'lin' variant."""
    import warnings
    out = {}
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        cosmo = cp.Cosmology(engine='eisenstein_hu')
        interp = cosmo.get_fourier().pk_interpolator()
        i1 = interp.to_1d(z=0.)
        xi1 = i1.to_xi()
        sq = np.geomspace(xi1.smin * 1.001, xi1.smax * 0.999, 96)
        kq = np.geomspace(1e-4, 10., 64)
        zq = np.array([0., 0.35, 1.1, 2.7])
        out['sq'], out['kq'], out['zq'] = sq, kq, zq
        out['xi1_s'], out['xi1_xi'] = np.asarray(xi1.s), np.asarray(xi1.xi)
        out['xi1_eval'] = xi1(sq)
        out['xi1_eval_oob'] = xi1(np.array([xi1.smin * 0.5, 1., xi1.smax * 2.]))
        pk1 = xi1.to_pk()
        out['xi1_to_pk_k'], out['xi1_to_pk_pk'] = np.asarray(pk1.k), np.asarray(pk1.pk)
        out['xi1_to_pk_eval'] = pk1(kq)
        out['xi1_sigma8'] = xi1.sigma8()
        out['xi1_sigma_r'] = xi1.sigma_r(np.array([2., 8., 30.]))
        out['xi1_sigma_d'] = xi1.sigma_d()
        xi2 = interp.to_xi()
        out['xi2_s'], out['xi2_z'] = np.asarray(xi2.s), np.asarray(xi2.z)
        out['xi2_eval'] = xi2(sq, zq)
        out['xi2_eval_nogrowth'] = xi2(sq, zq, ignore_growth=True)
        out['xi2_eval_pts'] = xi2(sq[:4], zq, grid=False)
        pk2 = xi2.to_pk()
        out['xi2_to_pk_eval'] = pk2(kq, zq)
        out['xi2_sigma8_z'] = xi2.sigma8_z(zq)
        out['xi2_to_1d_eval'] = xi2.to_1d(z=0.35)(sq)
        # the reference's own round trip (tests/test_interpolator.py:123-165) narrows the k range first: P(k) from the default
        # range has negative ringing at the edges, whose log makes the whole interpolator NaN (kept above: that IS the behaviour)
        c1 = i1.clone(extrap_kmin=1e-5, extrap_kmax=1e2)
        out['c1_k'], out['c1_pk'] = np.asarray(c1.k), np.asarray(c1.pk)
        xc1 = c1.to_xi()
        out['xc1_s'], out['xc1_xi'] = np.asarray(xc1.s), np.asarray(xc1.xi)
        pc1 = xc1.to_pk()
        out['xc1_to_pk_k'], out['xc1_to_pk_pk'] = np.asarray(pc1.k), np.asarray(pc1.pk)
        out['xc1_to_pk_eval'] = pc1(kq)
        out['xc1_sigma8'], out['xc1_sigma_d'] = xc1.sigma8(), xc1.sigma_d()
        out['xc1_sigma_r'] = xc1.sigma_r(np.array([2., 8., 30.]))
        c2 = interp.clone(extrap_kmin=1e-5, extrap_kmax=1e2)
        out['c2_k'], out['c2_z'], out['c2_pk'] = np.asarray(c2.k), np.asarray(c2.z), np.asarray(c2.pk)
        out['c2_eval'] = c2(kq, zq)
        xc2 = c2.to_xi()
        out['xc2_s'] = np.asarray(xc2.s)
        out['xc2_eval'] = xc2(sq, zq)
        pc2 = xc2.to_pk()
        out['xc2_to_pk_eval'] = pc2(kq, zq)
        out['xc2_sigma8_z'], out['xc2_sigma_dz'] = xc2.sigma8_z(zq), xc2.sigma_dz(zq)
        # Kirkby2013 correlation-function BAO filter on the 1D and 2D xi above: default boxes, rescaled boxes (rs_drag ratio != 1)
        from cosmoprimo.bao_filter import CorrelationFunctionBAOFilter
        f1 = CorrelationFunctionBAOFilter(xc1, engine='kirkby2013')
        out['kirkby_s'], out['kirkby1_xi'], out['kirkby1_xinow'] = np.asarray(f1.s), np.asarray(f1.xi), np.asarray(f1.xinow)
        out['kirkby1_smooth_eval'] = f1.smooth_xi_interpolator()(sq)
        other = cp.Cosmology(engine='eisenstein_hu', Omega_m=0.36, Omega_b=0.055, h=0.64, n_s=0.98, sigma8=0.85)
        f1r = CorrelationFunctionBAOFilter(xc1, engine='kirkby2013', cosmo=other, cosmo_fid=cosmo)
        out['kirkby1r_ratio'], out['kirkby1r_xinow'] = f1r.rs_drag_ratio(), np.asarray(f1r.xinow)
        f1d = CorrelationFunctionBAOFilter(xc1, engine='kirkby2013', cosmo=other)     # hard-coded fiducial rs_drag
        out['kirkby1d_ratio'], out['kirkby1d_xinow'] = f1d.rs_drag_ratio(), np.asarray(f1d.xinow)
        f2 = CorrelationFunctionBAOFilter(xc2, engine='kirkby2013', srange_left=(45., 80.), srange_right=(155., 195.), rescale_sbox=False, cosmo=other)
        out['kirkby2_xinow'] = np.asarray(f2.xinow)[:, ::6]
        out['kirkby2_smooth_eval'] = f2.smooth_xi_interpolator()(sq, zq)
        out['kirkby2_smooth_pk_eval'] = f2.smooth_pk_interpolator()(kq, zq)
        # tabulated: a (s, z) table of the same xi with its growth: 2D spline in both directions; and a 2-column 1D table
        st = np.geomspace(1e-2, 2e2, 300)
        zt = np.linspace(0., 2., 8)
        tab = xi2(st, zt)
        out['tab_s'], out['tab_z'], out['tab_xi'] = st, zt, tab
        t2 = cp.CorrelationFunctionInterpolator2D(st, zt, tab, interp_order_z=3)
        sq2 = np.geomspace(0.02, 150., 50)
        out['tab_sq'] = sq2
        out['tab2_eval'] = t2(sq2, np.array([0.1, 0.9, 1.7]))
        t1 = cp.CorrelationFunctionInterpolator1D(st, tab[:, :2])
        out['tab1_eval'] = t1(sq2)
        t1l = cp.CorrelationFunctionInterpolator1D(st, tab[:, 0], interp_s='lin')
        out['tab1_lin_eval'] = t1l(sq2)
    save('xi', **out)


def gen_bao2(cp):
    """f2: the remaining P(k) filters of the registry (hinton2017, savgol, ehsavgol, ehpoly, peakaverage) on the EH98 P(k) of the
    BAO_PARAMS cosmologies (1D callable input) and on a 4-column table; EH no-wiggle P(k) and peakaverage _prepare products."""
    import warnings
    out = {}
    engines = ['hinton2017', 'savgol', 'ehsavgol', 'ehpoly', 'peakaverage']
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        fid = cp.Cosmology(engine='eisenstein_hu')
        for i, par in enumerate(BAO_PARAMS):
            cosmo = cp.Cosmology(engine='eisenstein_hu', **par)
            interp = cosmo.get_fourier().pk_interpolator().to_1d(z=0.)
            for eng in engines:
                f = cp.PowerSpectrumBAOFilter(interp, engine=eng, cosmo=cosmo, cosmo_fid=fid)
                out['c%d_%s_pknow' % (i, eng)] = np.asarray(f.pknow)
            out['c%d_pk' % i] = np.asarray(f.pk)
            out['c%d_pknow_eh' % i] = cp.Fourier(cosmo, engine='eisenstein_hu_nowiggle', set_engine=False).pk_interpolator()(f.k, z=0.)
            out['c%d_rs_ratio' % i] = f.rs_drag_ratio()
        out['k'] = np.asarray(f.k)
        for j, kp in enumerate(f.k_peaks):
            out['peakaverage_k_peaks%d' % j] = np.asarray(kp)
            out['peakaverage_pad_peaks%d' % j] = np.asarray(f.pad_peaks[j])
        # tabulated 2D input (columns = 4 redshifts) of the third cosmology; no cosmo given to hinton2017 / savgol (rs ratio 1)
        cosmo = cp.Cosmology(engine='eisenstein_hu', **BAO_PARAMS[2])
        interp2 = cosmo.get_fourier().pk_interpolator()
        kt, zt = np.logspace(-5, 1.5, 400), np.array([0., 0.5, 1., 2.])
        tab = cp.PowerSpectrumInterpolator2D(kt, zt, interp2(kt, zt))
        out['tab_k'], out['tab_z'], out['tab_pk'] = kt, zt, interp2(kt, zt)
        for eng in engines:
            kw = dict(cosmo=cosmo, cosmo_fid=fid) if eng in ('ehsavgol', 'ehpoly', 'peakaverage') else {}
            out['tab_%s_pknow' % eng] = np.asarray(cp.PowerSpectrumBAOFilter(tab, engine=eng, **kw).pknow)
    save('bao2', **out)


def gen_bspline(cp):
    """The `bspline` P(k) filter (bao_filter.py:583-688).  Without constraint the reference runs as it is.  With constraints its mixing
    system ends in ``numpy.linalg.solve(system, target)`` with ``target`` one dimension short of ``system`` (:685): a stack of vectors for the
    numpy < 2 it was written for, a shape error under the numpy 2 of this image.  Those cases are generated with
    ``numpy.linalg.solve`` given back its numpy-1 reading of such a right-hand side for the duration of the call -- the reference's code is run
    unchanged -- and are stored under names ending in ``_np1``."""
    import warnings
    out = {}
    solve = np.linalg.solve

    def solve_np1(a, b):
        return solve(a, b[..., None])[..., 0] if np.ndim(b) == np.ndim(a) - 1 else solve(a, b)

    def run(interp, constraint, np1=None, **kw):
        np.linalg.solve = solve_np1 if (bool(constraint) if np1 is None else np1) else solve
        try:
            f = cp.PowerSpectrumBAOFilter(interp, engine='bspline', constraint=constraint, **kw)
            return f, np.asarray(f.pknow)
        finally:
            np.linalg.solve = solve

    cases = {'none': (), 'sigma8_np1': ('sigma8',), 'sigma8_sigmad_np1': ('sigma8', 'sigmad')}
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        for i, par in enumerate(BAO_PARAMS):
            cosmo = cp.Cosmology(engine='eisenstein_hu', **par)
            interp = cosmo.get_fourier().pk_interpolator().to_1d(z=0.)
            for name, constraint in cases.items():
                f, out['c%d_%s' % (i, name)] = run(interp, constraint, cosmo=cosmo)
            out['c%d_pk' % i] = np.asarray(f.pk)
            out['c%d_pknow_eh' % i] = cp.Fourier(cosmo, engine='eisenstein_hu_nowiggle', set_engine=False).pk_interpolator()(f.k, z=0.)
        out['k'] = np.asarray(f.k)
        cosmo = cp.Cosmology(engine='eisenstein_hu', **BAO_PARAMS[2])
        interp2 = cosmo.get_fourier().pk_interpolator()
        kt, zt = np.logspace(-5, 1.5, 400), np.array([0., 0.5, 1., 2.])
        tab = cp.PowerSpectrumInterpolator2D(kt, zt, interp2(kt, zt))
        out['tab_k'], out['tab_z'], out['tab_pk'] = kt, zt, interp2(kt, zt)
        for name, constraint in cases.items():      # several columns: the same shape error even without constraint
            out['tab_%s' % (name if name.endswith('_np1') else name + '_np1')] = run(tab, constraint, np1=True, cosmo=cosmo)[1]
    save('bspline', **out)


DENSITY_NAMES = ['rho_g', 'rho_b', 'rho_ur', 'rho_cdm', 'rho_k', 'rho_de', 'rho_Lambda', 'rho_fld', 'rho_r', 'rho_m', 'rho_tot', 'rho_crit',
                 'Omega_g', 'Omega_b', 'Omega_ur', 'Omega_cdm', 'Omega_k', 'Omega_de', 'Omega_Lambda', 'Omega_fld', 'Omega_r', 'Omega_m', 'T_cmb',
                 'rho_ncdm_tot', 'p_ncdm_tot']
RS_Z = [10., 1059.94, 1089.8]      # ba.rs(0.) raises in the reference (the fixed-depth Romberg rule misses its 1e-7 tolerance)
DENSITY_PARAMS = [dict(), dict(Omega_m=0.27, Omega_b=0.045, h=0.72, Omega_k=0.05), dict(Omega_m=0.36, h=0.64, w0_fld=-0.9, wa_fld=0.2, T_cmb=2.6),
                  dict(Omega_m=0.31, Omega_k=-0.03, w0_fld=-1.2, wa_fld=-0.4, N_ur=2.0328), dict(h=0.6766, w0_fld=-1., wa_fld=0., cs2_fld=0.9)]


def gen_densities(cp):
    """a19: every BaseBackground density / density parameter / T_cmb(z) for 5 cosmologies (flat LCDM, curved, w0-wa, cs2_fld != 1)."""
    import warnings
    z = np.concatenate([[0., 1e-3], np.linspace(0.05, 3., 10), [10., 100., 1100., 9999.]])
    out = {'z': z}
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        for i, par in enumerate(DENSITY_PARAMS):
            ba = cp.Cosmology(engine='eisenstein_hu', **par).get_background()
            for name in DENSITY_NAMES:
                out['c%d_%s' % (i, name)] = np.asarray(getattr(ba, name)(z), dtype='f8') + 0. * z
            out['c%d_has_fld' % i] = float(ba.Omega0_fld != 0.)
            out['c%d_rs' % i] = np.array([float(ba.rs(zz)) for zz in RS_Z])
            cosmo_i = cp.Cosmology(engine='eisenstein_hu', **par)
            out['c%d_theta_cosmomc' % i] = float(cosmo_i['theta_cosmomc'])
            from cosmoprimo.cosmology import DefaultBackground
            bd = DefaultBackground(cosmo_i.engine if hasattr(cosmo_i, 'engine') else cosmo_i._engine)
            zg = np.concatenate([[0., 1e-3], np.linspace(0.05, 3., 10), [10., 100., 400.]])
            out['zg'] = zg
            out['c%d_growth_factor_ode' % i] = np.asarray(bd.growth_factor(zg))
            out['c%d_growth_factor_ode_znorm' % i] = np.asarray(bd.growth_factor(zg, znorm=10.))
            out['c%d_growth_factor_ode_cb' % i] = np.asarray(bd.growth_factor(zg, mass='cb'))
            out['c%d_growth_rate_ode' % i] = np.asarray(bd.growth_rate(zg))
            out['c%d_time' % i] = np.asarray(ba.time(z), dtype='f8')
            out['c%d_age' % i] = float(ba.age)
        out['time_knots'] = np.asarray(ba._cache['time']._x)
        out['time_nan_outside'] = ba.time(np.array([-0.1, 1e8]))
    save('densities', **out)


NCDM_PARAMS = [dict(m_ncdm=0.06), dict(m_ncdm=[0.02, 0.03, 0.05], Omega_m=0.31, h=0.6766, w0_fld=-0.9, wa_fld=0.15),
               dict(m_ncdm=[0.1, 0.4], Omega_k=0.02, T_cmb=2.6, N_eff=3.2), dict(m_ncdm=[0.06], T_ncdm_over_cmb=[0.7], Omega_cdm=0.26, N_ur=2.03)]


def gen_ncdm(cp):
    """a23: massive neutrinos: derived parameters, interpolated rho / p of every species, E(z), densities and density parameters,
    distances and time for 4 cosmologies (one / three / two species; w0wa; curvature; explicit T_ncdm_over_cmb and N_ur)."""
    import warnings
    z = np.concatenate([[0., 1e-3], np.linspace(0.05, 3., 12), [10., 100., 1100., 9999.]])
    out = {'z': z}
    names = ['efunc', 'rho_ncdm_tot', 'p_ncdm_tot', 'rho_m', 'rho_r', 'rho_tot', 'rho_crit', 'Omega_m', 'Omega_r', 'Omega_ncdm_tot', 'Omega_pncdm_tot',
             'Omega_de', 'comoving_radial_distance', 'angular_diameter_distance', 'luminosity_distance', 'time']
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        for i, par in enumerate(NCDM_PARAMS):
            cosmo = cp.Cosmology(engine='eisenstein_hu', **par)
            ba = cosmo.get_background()
            for name in ['N_ur', 'N_eff', 'Omega_ncdm_tot', 'Omega_pncdm_tot', 'Omega_m', 'Omega_de', 'Omega_cdm', 'Omega_r', 'm_ncdm_tot']:
                out['c%d_par_%s' % (i, name)] = float(cosmo[name])
            out['c%d_par_Omega_ncdm' % i] = np.asarray(cosmo['Omega_ncdm'], dtype='f8')
            out['c%d_par_T_ncdm' % i] = np.asarray(cosmo['T_ncdm'], dtype='f8')
            for name in names:
                out['c%d_%s' % (i, name)] = np.asarray(getattr(ba, name)(z), dtype='f8')
            out['c%d_rho_ncdm' % i] = np.asarray(ba.rho_ncdm(z), dtype='f8')
            out['c%d_p_ncdm' % i] = np.asarray(ba.p_ncdm(z), dtype='f8')
            out['c%d_Omega_ncdm_z' % i] = np.asarray(ba.Omega_ncdm(z), dtype='f8')
            out['c%d_T_ncdm_z' % i] = np.asarray(ba.T_ncdm(z), dtype='f8')
            out['c%d_age' % i] = float(ba.age)
        out['ncdm_knots'] = np.asarray(ba._cache['rho_ncdm']._x)
        out['rho_ncdm_table'] = np.asarray(ba._cache['rho_ncdm']._fun)
    save('ncdm', **out)


VARIANTS_PARAMS = [dict(), dict(m_ncdm=0.06), dict(m_ncdm=[0.02, 0.03, 0.05], Omega_m=0.31, h=0.6766, n_s=0.9665, sigma8=0.81),
                   dict(m_ncdm=[0.1, 0.4], Omega_b=0.045, T_cmb=2.6, A_s=2.2e-9)]


def gen_variants(cp):
    """f3: eisenstein_hu_nowiggle_variants: engine scalars, transfer_kz (delta_m, delta_cb), P(k, z), sigma8 and the rescaling factor for 4
    cosmologies without / with one / three / two massive species."""
    import warnings
    k = np.concatenate([np.logspace(-5, 2, 36), [0.05, 0.1, 1.]])
    z = np.array([0., 0.5, 1.5, 4.])
    out = {'k': k, 'z': z}
    names = ['omega_b', 'omega_m', 'frac_b', 'frac_cdm', 'frac_cb', 'frac_ncdm', 'theta_cmb', 'z_eq', 'k_eq', 'z_drag', 'rs_drag', 'p_c', 'p_cb',
             'gamma_ncdm', 'beta_c']
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        for i, par in enumerate(VARIANTS_PARAMS):
            cosmo = cp.Cosmology(engine='eisenstein_hu_nowiggle_variants', **par)
            eng = cosmo._engine if hasattr(cosmo, '_engine') else cosmo.engine
            for name in names:
                out['c%d_%s' % (i, name)] = float(getattr(eng, name))
            tr, fo, ba = cosmo.get_transfer(), cosmo.get_fourier(), cosmo.get_background()
            out['c%d_growth_k0' % i] = np.asarray(ba.growth_factor(z, znorm=eng.z_eq))
            for of in ['delta_m', 'delta_cb']:
                out['c%d_transfer_%s' % (i, of)] = np.asarray(tr.transfer_kz(k, z, of=of))
                out['c%d_pk_%s' % (i, of)] = np.asarray(fo.pk_interpolator(of=of)(k, z))
            out['c%d_pk_theta' % i] = np.asarray(fo.pk_interpolator(of='theta_m')(k, z))
            out['c%d_sigma8_m' % i] = float(fo.sigma8_m)
            out['c%d_sigma8_z' % i] = np.asarray(fo.sigma8_z(z))
            out['c%d_rsigma8' % i] = float(eng._rsigma8)
            out['c%d_A_s' % i] = float(cosmo.get_primordial().A_s)
            out['c%d_rs_drag_th' % i] = float(cosmo.get_thermodynamics().rs_drag)
    save('variants', **out)


# massive neutrinos through the analytic engines (the reference computes: eisenstein_hu.py:21-33, its warnings are commented out): 'desi' is
# fiducial.DESI(engine=...) (m_ncdm = [0.06], the AbacusSummit base cosmology); the others as the variants / ncdm targets take them
POWER_NCDM_CASES = [dict(m_ncdm=[0.06]), 'desi', dict(m_ncdm=[0.02, 0.03, 0.05], Omega_m=0.31, h=0.6766, n_s=0.9665, sigma8=0.81, w0_fld=-0.9, wa_fld=0.1),
                    dict(m_ncdm=[0.1, 0.4], Omega_b=0.045, T_cmb=2.6, A_s=2.2e-9, Omega_k=0.01), dict(m_ncdm=0.3, neutrino_hierarchy='normal', Omega_m=0.29)]
POWER_NCDM_FILTERS = ['wallish2018', 'brieden2022', 'ehpoly', 'hinton2017', 'peakaverage', 'savgol', 'ehsavgol']


def _power_ncdm_cosmo(cp, case, engine):
    if case == 'desi':
        from cosmoprimo.fiducial import DESI
        return DESI(engine=engine)
    return cp.Cosmology(engine=engine, **case)


def gen_power_ncdm(cp):
    """a29 with massive species: the three analytic engines on POWER_NCDM_CASES -- engine scalars, rs_drag / z_drag, transfer_k, pk_k, P(k, z) of
    delta_m and theta_m, sigma8_z, sigma8_m, growth factor / rate, the sigma8 rescaling factor and A_s; the cosmology's derived Omega_m / Omega_cdm;
    and the P(k) filters with cosmo = cosmo_fid = DESI(engine='eisenstein_hu') and cosmo = the first case (rs_drag ratio != 1)."""
    import warnings
    k = np.concatenate([np.logspace(-7, 2, 46), [1e-5, 0.05, 0.1, 1.]])
    z = np.array([0., 0.5, 1., 2., 3.])
    out = {'k': k, 'z': z}
    names_sc = ['z_eq', 'k_eq', 'z_drag', 'r_drag', 'r_eq', 'rs_drag', 'k_silk', 'alpha_c', 'beta_c', 'alpha_b', 'beta_node', 'beta_b', 'alpha_gamma', 'gamma']
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        for eng in ['eisenstein_hu', 'eisenstein_hu_nowiggle', 'bbks']:
            for i, case in enumerate(POWER_NCDM_CASES):
                cosmo = _power_ncdm_cosmo(cp, case, eng)
                fo, tr, pm, ba = cosmo.get_fourier(), cosmo.get_transfer(), cosmo.get_primordial(), cosmo.get_background()
                pre = '%s_c%d_' % (eng, i)
                out[pre + 'transfer'] = np.asarray(tr.transfer_k(k))
                out[pre + 'pk_prim'] = np.asarray(pm.pk_k(k))
                out[pre + 'pkz'] = np.asarray(fo.pk_interpolator()(k, z))
                out[pre + 'pkz_theta'] = np.asarray(fo.pk_interpolator(of='theta_m')(k, z))
                out[pre + 'pkz_delta_theta'] = np.asarray(fo.pk_interpolator(of=('delta_m', 'theta_m'))(k, z))
                out[pre + 'sigma8_z'] = np.asarray(fo.sigma8_z(z))
                out[pre + 'sigma_rz'] = np.asarray(fo.sigma_rz(np.array([2., 8., 30.]), z))
                out[pre + 'sigma8_m'] = float(fo.sigma8_m)
                out[pre + 'growth_factor'] = np.asarray(ba.growth_factor(z))
                out[pre + 'growth_factor_znorm0'] = np.asarray(ba.growth_factor(z, znorm=0.))
                out[pre + 'growth_rate'] = np.asarray(ba.growth_rate(z))
                out[pre + 'rsigma8'] = float(cosmo._engine._rsigma8)
                out[pre + 'A_s'] = float(pm.A_s)
                for name in ['Omega_m', 'Omega_cdm', 'Omega_de', 'Omega_ncdm_tot', 'Omega_pncdm_tot', 'N_ur', 'h', 'Omega_b', 'Omega_k', 'T_cmb', 'w0_fld', 'wa_fld',
                             'n_s', 'alpha_s', 'beta_s', 'k_pivot']:
                    out[pre + 'par_' + name] = float(cosmo[name])
                out[pre + 'par_m_ncdm'] = np.asarray(cosmo['m_ncdm'], dtype='f8')
                out[pre + 'par_T_ncdm_over_cmb'] = np.asarray(cosmo['T_ncdm_over_cmb'], dtype='f8')
                out[pre + 'A_s_fid'] = float(cosmo._engine._A_s)      # the amplitude before the sigma8 rescaling (cosmology.py:505-510)
                for name in names_sc:
                    v = getattr(cosmo._engine, name, None)
                    if v is not None:
                        out[pre + name] = float(v)
                if eng != 'bbks':
                    th = cosmo.get_thermodynamics()
                    out[pre + 'rs_drag_h'], out[pre + 'z_drag_th'] = float(th.rs_drag), float(th.z_drag)
        # the filters on cosmologies with a massive species: cosmo_fid = DESI(); cosmo = DESI() (ratio 1) and cosmo = the first case
        fid = _power_ncdm_cosmo(cp, 'desi', 'eisenstein_hu')
        for j, case in enumerate(['desi', POWER_NCDM_CASES[0], POWER_NCDM_CASES[2]]):
            cosmo = _power_ncdm_cosmo(cp, case, 'eisenstein_hu')
            interp = cosmo.get_fourier().pk_interpolator().to_1d(z=0.)
            for name in POWER_NCDM_FILTERS:
                f = cp.PowerSpectrumBAOFilter(interp, engine=name, cosmo=cosmo, cosmo_fid=fid)
                out['filter%d_%s_pknow' % (j, name)] = np.asarray(f.pknow)
            out['filter%d_pk' % j] = np.asarray(f.pk)
            out['filter%d_rs_ratio' % j] = float(f.rs_drag_ratio())
            out['filter%d_pknow_eh' % j] = np.asarray(cp.Fourier(cosmo, engine='eisenstein_hu_nowiggle', set_engine=False).pk_interpolator()(f.k, z=0.))
            interp2d = cosmo.get_fourier().pk_interpolator(z=np.array([0.]))
            for name in ['wallish2018', 'brieden2022']:
                out['filter%d_%s_pknow_2d' % (j, name)] = np.asarray(cp.PowerSpectrumBAOFilter(interp2d, engine=name, cosmo=cosmo, cosmo_fid=fid).pknow)
        out['filter_k'] = np.asarray(f.k)
    save('power_ncdm', **out)


BAO_BATCH_FILTERS = ['peakaverage', 'ehpoly', 'hinton2017', 'ehsavgol', 'savgol']
BAO_BATCH_STRIDE = 4


def bao_batch_params(n=24, seed=7):
    """The cosmologies of the 'bao_batch' target: rs_drag ratios to the default fiducial cosmology from 0.90 to 1.12."""
    rng = np.random.RandomState(seed)
    return dict(Omega_m=rng.uniform(0.24, 0.40, n), Omega_b=rng.uniform(0.04, 0.06, n), h=rng.uniform(0.6, 0.8, n), n_s=rng.uniform(0.92, 1., n))


def gen_bao_batch(cp):
    """f2 over batches: the P(k) filters that depend on the cosmology through its rs_drag ratio and its no-wiggle template (peakaverage, ehpoly, ehsavgol)
    or on the spectrum's own maximum (hinton2017), cosmology by cosmology through the reference for 24 cosmologies (every 4th wavenumber kept): what a
    batched run must reproduce entry by entry."""
    import warnings
    par = bao_batch_params()
    n = len(par['h'])
    out = {}
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        fid = cp.Cosmology(engine='eisenstein_hu')
        acc = {name: [] for name in BAO_BATCH_FILTERS}
        ratios = []
        for i in range(n):
            cosmo = cp.Cosmology(engine='eisenstein_hu', sigma8=0.8, **{name: float(v[i]) for name, v in par.items()})
            interp = cosmo.get_fourier().pk_interpolator(z=np.array([0.]))
            for name in BAO_BATCH_FILTERS:
                f = cp.PowerSpectrumBAOFilter(interp, engine=name, cosmo=cosmo, cosmo_fid=fid)
                acc[name].append(np.asarray(f.pknow)[::BAO_BATCH_STRIDE, 0])
            ratios.append(float(f.rs_drag_ratio()))
        for name in BAO_BATCH_FILTERS:
            out[name] = np.array(acc[name])
        out['rs_ratio'] = np.array(ratios)
        out['k'] = np.asarray(f.k)[::BAO_BATCH_STRIDE]
    save('bao_batch', **out)


CALCULATOR_CASES = [('eisenstein_hu', {}, dict(Omega_m=0.3)), ('eisenstein_hu', {}, dict(h=0.65, n_s=0.95, w0_fld=-0.9)),
                    ('eisenstein_hu_nowiggle_variants', dict(m_ncdm=[0.06]), dict(Omega_m=0.28)), ('bbks', {}, dict(Omega_b=0.045))]
CALCULATOR_PK_STRIDE = (8, 3)


def gen_calculator(cp):
    """The reference's batch driver (emulators/__init__.py:11-60) on its analytic engines; its calculator drops the power spectra of these
    engines (swallowed keyword error), so the pairs are taken from ``Fourier.pk_interpolator`` directly, on a sub-grid of the default (k, z)."""
    from cosmoprimo.emulators import get_calculator
    from cosmoprimo.emulators.emulated import get_default_k_callable, get_default_z_callable
    out = {}
    k, z = get_default_k_callable()[::CALCULATOR_PK_STRIDE[0]], get_default_z_callable()[::CALCULATOR_PK_STRIDE[1]]
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        for i, (engine, base, params) in enumerate(CALCULATOR_CASES):
            cosmo = cp.Cosmology(engine=engine, **base)
            for name, value in get_calculator(cosmo)(**params).items():
                out['c%d:%s' % (i, name)] = np.asarray(value, dtype='f8')
            fo = cosmo.clone(**params).get_fourier()
            for of in ([('delta_m', 'delta_m'), ('delta_m', 'theta_m'), ('theta_m', 'theta_m')] if 'variants' not in engine else [('delta_m', 'delta_m')]):
                out['c%d:fourier.pk.%s.%s' % ((i,) + of)] = np.asarray(fo.pk_interpolator(of=of)(k, z))
    save('calculator', **out)


def gen_cosmology_api(cp):
    """Host-side behaviour of ``Cosmology`` the reference's tests/test_cosmology.py exercises: default-parameter names, ``solve``, clones."""
    from cosmoprimo.fiducial import DESI
    out = {}
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        out['default_names'] = np.array(sorted(cp.Cosmology.get_default_params()))
        out['default_names_noconflicts'] = np.array(sorted(cp.Cosmology.get_default_params(include_conflicts=False)))
        out['default_cosmology_names'] = np.array(sorted(cp.Cosmology.get_default_params(of='cosmology')))
        cosmo = cp.Cosmology(engine='eisenstein_hu')
        out['z_pk'] = np.asarray(cosmo['z_pk'])
        solved = cosmo.solve('h', 'theta_MC_100', 1.04092)
        out['solve_h_theta_MC_100'] = float(solved['h'])
        solved = DESI(engine='eisenstein_hu_nowiggle_variants').solve('h', lambda cosmo: 100. * cosmo['theta_cosmomc'], target=1.04, limits=[0.6, 0.9], xtol=1e-6)
        out['solve_h_desi'] = float(solved['h'])
        cosmo = cp.Cosmology(omega_cdm=0.2, engine='eisenstein_hu')
        clone = cosmo.clone(base='internal', h=cosmo.h * 1.1)
        out['clone_internal'] = np.array([clone.Omega0_m, clone.Omega0_cdm, clone['omega_cdm']], dtype='f8')
        clone = cosmo.clone(base='input', h=cosmo.h * 1.1)
        out['clone_input'] = np.array([clone.Omega0_m, clone.Omega0_cdm, clone['omega_cdm']], dtype='f8')
        cosmo = cp.Cosmology(Omega_g=5e-5, omega_ur=1.7e-5)
        out['from_Omega_g_ur'] = np.array([cosmo['T_cmb'], cosmo['N_ur'], cosmo['N_eff']], dtype='f8')
        cosmo = cp.Cosmology(r=0.1)
        out['tensor_defaults'] = np.array([cosmo['n_t'], cosmo['alpha_t']], dtype='f8')
    save('cosmology_api', **out)


def gen_abacus_table(cp):
    """The AbacusSummit cosmology table (https://github.com/abacusorg/AbacusSummit/tree/master/Cosmologies; the reference carries it as
    data/abacus_cosmologies.csv) as read by the reference's ``AbacusSummit_params``: package data, cosmoprimo_amd/data/abacus_cosmologies.json."""
    import json
    from cosmoprimo.fiducial import AbacusSummit_params
    names = ['root', 'omega_b', 'omega_cdm', 'h', 'A_s', 'n_s', 'alpha_s', 'N_ur', 'omega_ncdm', 'w0_fld', 'wa_fld']
    table = {}
    for row in AbacusSummit_params(params=names):
        root = row.pop('root')
        table[root] = {name: (list(value) if isinstance(value, tuple) else value) for name, value in row.items()}
    fn = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'cosmoprimo_amd', 'data', 'abacus_cosmologies.json')
    os.makedirs(os.path.dirname(fn), exist_ok=True)
    with open(fn, 'w') as file:
        json.dump(table, file, indent=0)
    print('wrote', fn, len(table), 'cosmologies')
    # and what the reference derives for a few of them, as a fixture
    out = {}
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        from cosmoprimo.fiducial import AbacusSummit
        for name in ['000', '009', '019', '130']:
            cosmo = AbacusSummit(name, engine='eisenstein_hu_nowiggle_variants')
            out['c%s' % name] = np.array([cosmo['h'], cosmo['Omega_m'], cosmo['N_ur'], cosmo['N_eff'], cosmo['m_ncdm_tot'], cosmo['N_ncdm'], cosmo['w0_fld'],
                                          cosmo['wa_fld'], cosmo.get_primordial().A_s, cosmo.comoving_radial_distance(1.)], dtype='f8')
    save('abacus', **out)


def gen_desi_table():
    """161 of the 40 002 rows of the reference's tabulated DESI fiducial (cosmoprimo/data/desi.dat: z, E(z), D_C(z) [Mpc/h], computed with
    a Boltzmann code): a data file of the reference kept as a fixture, the z = 0 row plus 160 rows evenly spaced in log z."""
    table = np.loadtxt(os.path.join(REFERENCE_ROOT, "cosmoprimo", "data", "desi.dat"), comments='#')
    idx = np.r_[0, np.linspace(1, len(table) - 1, 160).astype(int)]
    save('desi_table', z=table[idx, 0], efunc=table[idx, 1], comoving_radial_distance=table[idx, 2])


def gen_api_flows(cp):
    """tests/api_scenarios.py (API scenarios written for this repository) replayed with the reference: every entry is what the reference
    returns -- an array, or the class name of the exception it raises."""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests'))
    import api_scenarios
    results = api_scenarios.run_all(cp)
    save('api_flows', **results)


# Cosmologies drawn from wide priors -- curvature, (w0, wa), one to three massive species, N_eff, T_cmb, sigma8 or A_s -- through the reference's analytic
# engines and background: the corners the hand-picked cases of the other targets do not visit.
FUZZ_N = 48
FUZZ_ENGINES = ['eisenstein_hu', 'eisenstein_hu_nowiggle', 'bbks', 'eisenstein_hu_nowiggle_variants']


def fuzz_params(n=FUZZ_N, seed=20261004):
    """The n parameter sets (dicts of plain floats / lists): what the generator and tests/test_fuzz_gpu.py both build their cosmologies from."""
    rng = np.random.default_rng(seed)
    cases = []
    species = [None, [0.06], [0.1, 0.3], [0.02, 0.05, 0.2]]
    for i in range(n):
        par = dict(h=rng.uniform(0.5, 0.9), Omega_cdm=rng.uniform(0.15, 0.45), Omega_b=rng.uniform(0.03, 0.07), n_s=rng.uniform(0.85, 1.05),
                   T_cmb=rng.uniform(2.6, 2.9), N_eff=rng.uniform(2.6, 4.))
        if i % 2:
            par['Omega_k'] = rng.uniform(-0.15, 0.15)
        if i % 3:
            par['w0_fld'], par['wa_fld'] = rng.uniform(-1.6, -0.5), rng.uniform(-1., 0.4)
        if i % 4 < 2:
            par['sigma8'] = rng.uniform(0.6, 1.)
        else:
            par['A_s'] = rng.uniform(1.5e-9, 3e-9)
        m = species[(i // 2) % 4]
        if m is not None:
            par['m_ncdm'] = [float(x * rng.uniform(0.8, 1.2)) for x in m]
        cases.append({name: (value if isinstance(value, list) else float(value)) for name, value in par.items()})
    return cases


def gen_fuzz(cp):
    """FUZZ_N random cosmologies x the four analytic engines: P(k, z), sigma8_z, growth factor / rate, rs_drag / z_drag, rescaling factor; the
    background of each (efunc, distances, time, age, Omega_m(z), ...) and its compiled parameters; wallish2018 and brieden2022 of every fourth.
    Stacked over the cosmologies: <engine>_<quantity> (FUZZ_N, ...), <quantity> (FUZZ_N, ...); m_ncdm padded to three species with NaN."""
    import warnings
    k = np.geomspace(1e-4, 10., 12)
    z = np.array([0., 0.8, 2.5])
    zb = np.array([0.01, 0.5, 1.5, 3., 10.])
    rows = {}

    def put(name, value):
        rows.setdefault(name, []).append(np.asarray(value, dtype='f8'))

    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        fid = cp.Cosmology(engine='eisenstein_hu')
        for i, par in enumerate(fuzz_params()):
            for eng in FUZZ_ENGINES:
                cosmo = cp.Cosmology(engine=eng, **par)
                fo, ba = cosmo.get_fourier(), cosmo.get_background()
                put(eng + '_pkz', fo.pk_interpolator()(k, z))
                put(eng + '_sigma8_z', fo.sigma8_z(z))
                put(eng + '_growth_factor', ba.growth_factor(z))
                put(eng + '_growth_rate', ba.growth_rate(z))
                if eng != 'bbks':
                    th = cosmo.get_thermodynamics()
                    put(eng + '_rs_drag', th.rs_drag)
                    put(eng + '_z_drag', th.z_drag)
                if eng != 'eisenstein_hu_nowiggle_variants':
                    put(eng + '_rsigma8', cosmo._engine._rsigma8)
                    put(eng + '_A_s_fid', cosmo._engine._A_s)
            # (the background is the engines' common DefaultBackground: taken from the last one); the compiled parameters, for the oracle
            for name in ['efunc', 'comoving_radial_distance', 'angular_diameter_distance', 'luminosity_distance', 'time', 'Omega_m', 'Omega_de', 'rho_ncdm_tot']:
                put(name, getattr(ba, name)(zb))
            put('age', ba.age)
            for name in ['h', 'Omega_cdm', 'Omega_b', 'Omega_k', 'T_cmb', 'N_ur', 'w0_fld', 'wa_fld', 'n_s', 'alpha_s', 'beta_s', 'k_pivot', 'Omega_m', 'Omega_de']:
                put('par_' + name, cosmo[name])
            for name in ['m_ncdm', 'T_ncdm_over_cmb']:
                v = np.asarray(cosmo[name], dtype='f8').ravel()
                put('par_' + name, np.concatenate([v, np.full(3 - v.size, np.nan)]))
            if i % 4 == 0:
                cosmo = cp.Cosmology(engine='eisenstein_hu', **par)
                interp = cosmo.get_fourier().pk_interpolator().to_1d(z=0.)
                for name in ['wallish2018', 'brieden2022']:
                    put(name, np.asarray(cp.PowerSpectrumBAOFilter(interp, engine=name, cosmo=cosmo, cosmo_fid=fid).pknow)[::8])
    save('fuzz', k=k, z=z, zb=zb, **{name: np.stack(v) for name, v in rows.items()})


# FFTLog configurations drawn at random: class, size (powers of two and not), range, tilt, folds, low-ringing, xy, padding mode, batch shape -- the
# combinations the hand-picked G3 cases do not visit.  A configuration is a plain dict (tests rebuild the same object from it).
FFTLOG_FUZZ_N = 60


def fftlog_fuzz_configs(n=FFTLOG_FUZZ_N, seed=20261005):
    rng = np.random.default_rng(seed)
    kinds = ['PowerToCorrelation', 'CorrelationToPower', 'TophatVariance', 'GaussianVariance', 'HankelTransform', 'FFTlog']
    extraps = [0, 'log', 'edge', 0.7, ('log', 0.), (1.5, 'edge'), ('edge', 'log')]
    configs = []
    for i in range(n):
        kind = kinds[i % len(kinds)]
        cfg = dict(kind=kind, n=int(rng.choice([64, 100, 127, 256, 500, 1024, 2048, 3000])), lo=float(rng.uniform(-6., -2.)), span=float(rng.uniform(4., 9.)),
                   q=float(np.round(rng.uniform(-0.4, 0.9), 3)), minfolds=int(rng.choice([1, 2, 3])), lowring=bool(rng.integers(2)),
                   xy=float(rng.choice([0.3, 1., 4.])), extrap=extraps[int(rng.integers(len(extraps)))], keep_padding=bool(i % 7 == 3),
                   nbatch=int(rng.choice([0, 0, 3])), slope=float(rng.uniform(-2.5, 1.5)), knee=float(rng.uniform(0.2, 0.8)))
        if kind in ('PowerToCorrelation', 'CorrelationToPower'):
            cfg['ell'] = [0, 2, 4] if i % 5 == 0 else int(rng.choice([0, 1, 2, 3, 4]))
            cfg['complex'] = bool(kind == 'PowerToCorrelation' and i % 4 == 1)
        if kind == 'HankelTransform':
            cfg['nu'] = float(rng.choice([0., 0.5, 1., 2.5]))
        if kind == 'FFTlog':
            cfg['kernel'] = [('SphericalBesselJKernel', 1), ('TophatKernel', 3), ('TophatSqKernel', 1), ('TophatSqKernel', 2), ('GaussianKernel', None),
                             ('BesselJKernel', 1.5)][int(rng.integers(6))]
        configs.append(cfg)
    return configs


def fftlog_fuzz_stride(size):
    """Every how many-th output sample is kept in the fixture (at most 128 per row)."""
    return max(1, -(-size // 128))


def fftlog_fuzz_error(cfg, got, ref, y):
    """max |got - ref| y^q / max |ref| y^q per row: norm-wise in the tilted space, where the transform's rounding is uniform."""
    q = cfg['q'] + (1.5 if cfg['kind'] in ('PowerToCorrelation', 'CorrelationToPower', 'TophatVariance', 'GaussianVariance') else 0.)
    tilt = np.broadcast_to(y, ref.shape[-y.ndim:] if y.ndim > 1 else ref.shape[-1:])**q
    scale = np.abs(ref * tilt).max(axis=-1, keepdims=True)
    return float((np.abs((got - ref) * tilt) / scale).max())


def fftlog_fuzz_build(fl, cfg):
    """(transform object of the module ``fl`` -- the reference's fftlog or this package's --, x, input function) of a configuration."""
    x = np.logspace(cfg['lo'], cfg['lo'] + cfg['span'], cfg['n'])
    kw = dict(q=cfg['q'], minfolds=cfg['minfolds'], lowring=cfg['lowring'], xy=cfg['xy'])
    kind = cfg['kind']
    if kind in ('PowerToCorrelation', 'CorrelationToPower'):
        obj = getattr(fl, kind)(x, ell=cfg['ell'], **(dict(complex=True) if cfg['complex'] else {}), **kw)
    elif kind == 'HankelTransform':
        obj = fl.HankelTransform(x, nu=cfg['nu'], **kw)
    elif kind == 'FFTlog':
        name, arg = cfg['kernel']
        obj = fl.FFTlog(x, getattr(fl, name)(*(() if arg is None else (arg,))), **kw)
    else:
        obj = getattr(fl, kind)(x, **kw)
    # a smooth positive function of x: a power law with a knee inside the range (positive: the log extrapolation takes it)
    xm = 10.**(cfg['lo'] + cfg['knee'] * cfg['span'])
    fun = (x / xm)**cfg['slope'] / (1. + (x / xm)**2)**1.5
    if cfg['nbatch']:
        fun = fun * np.array([1., 0.5, 2.5])[:cfg['nbatch'], None] * (x / xm)**(0.1 * np.arange(cfg['nbatch'])[:, None])
        if np.ndim(cfg.get('ell', 0)):      # several transforms at once: (batch, nell, n)
            fun = fun[:, None, :] * np.ones((1, len(cfg['ell']), 1))
    return obj, x, fun


def fftlog_large_configs():
    """Sizes beyond the LDS-resident kernel (padded length 16 384 ... 131 072: the four-step path, cp_fftlog_large.hip), a few of each class."""
    configs = fftlog_fuzz_configs(n=12, seed=20261010)
    sizes = [6000, 10000, 20000, 40000]
    for i, cfg in enumerate(configs):
        cfg['n'] = sizes[i % 4]
        cfg['minfolds'] = 2 if i % 3 else 3
        cfg['nbatch'] = 3 if i % 4 == 1 else 0
        if np.ndim(cfg.get('ell', 0)):
            cfg['ell'] = [0, 2]
    return configs


def gen_fftlog_large(cp):
    """The same record as fftlog_fuzz (transforms strided to 128 samples, the reference's own movement under one-ulp inputs) for fftlog_large_configs()."""
    from cosmoprimo import fftlog as fl
    out = {}
    for i, cfg in enumerate(fftlog_large_configs()):
        obj, x, fun = fftlog_fuzz_build(fl, cfg)
        y, g = obj(fun, extrap=cfg['extrap'], keep_padding=cfg['keep_padding'])
        y, g = np.asarray(y), np.asarray(g)
        stride = fftlog_fuzz_stride(y.shape[-1])
        out['c%d_y' % i], out['c%d_g' % i], out['c%d_size' % i] = y[..., ::stride], g[..., ::stride], np.array(y.shape[-1])
        pre = obj.padded_prefactor
        up = np.random.default_rng(2000 + i).integers(2, size=np.shape(pre)).astype(bool)
        obj.padded_prefactor = np.where(up, np.nextafter(pre, np.inf), np.nextafter(pre, -np.inf))
        moved = np.asarray(obj(fun, extrap=cfg['extrap'], keep_padding=cfg['keep_padding'])[1])
        obj.padded_prefactor = pre
        out['c%d_moves' % i] = np.array(fftlog_fuzz_error(cfg, moved, g, y))
    save('fftlog_large', **out)


def gen_fftlog_fuzz(cp):
    """FFTLOG_FUZZ_N random FFTLog configurations through the reference: output coordinates and transforms, every fftlog_fuzz_stride-th sample
    of them: keys c<i>_y / c<i>_g (complex as it comes), c<i>_size the full length."""
    from cosmoprimo import fftlog as fl
    out = {}
    for i, cfg in enumerate(fftlog_fuzz_configs()):
        obj, x, fun = fftlog_fuzz_build(fl, cfg)
        y, g = obj(fun, extrap=cfg['extrap'], keep_padding=cfg['keep_padding'])
        y, g = np.asarray(y), np.asarray(g)
        stride = fftlog_fuzz_stride(y.shape[-1])
        out['c%d_y' % i], out['c%d_g' % i], out['c%d_size' % i] = y[..., ::stride], g[..., ::stride], np.array(y.shape[-1])
        # how far the reference's own result moves when every input sample is off by one rounding error: with constant / edge padding over many
        # decades the cropped output is orders of magnitude below what the padded transform carries, and holds that many digits fewer -- the
        # tolerance a parity test can ask for (tests add 30 x this to their 1e-12)
        pre = obj.padded_prefactor
        up = np.random.default_rng(1000 + i).integers(2, size=np.shape(pre)).astype(bool)
        obj.padded_prefactor = np.where(up, np.nextafter(pre, np.inf), np.nextafter(pre, -np.inf))      # (every padded, prefactored sample one ulp off)
        moved = np.asarray(obj(fun, extrap=cfg['extrap'], keep_padding=cfg['keep_padding'])[1])
        obj.padded_prefactor = pre
        out['c%d_moves' % i] = np.array(fftlog_fuzz_error(cfg, moved, g, y))
    save('fftlog_fuzz', **out)


# Tabulated P(k) / P(k, z) interpolators with options drawn at random -- grid sizes and spacings (geometric, jittered), interpolation in k or log k,
# extrapolation mode and range, spline degrees, growth factor or table in z -- evaluated inside, in the extrapolation range and outside, with the sigma
# integrals and the xi side: the option combinations the hand-picked G4 / G5 cases do not visit.
INTERP_FUZZ_N = 36


def interp_fuzz_configs(n=INTERP_FUZZ_N, seed=20261006):
    rng = np.random.default_rng(seed)
    configs = []
    for i in range(n):
        two_d = bool(i % 2)
        log_k = bool(i % 5 != 3)
        cfg = dict(two_d=two_d, nk=int(rng.choice([24, 57, 200, 540])), kmin=float(10.**rng.uniform(-4.5, -3.)), kmax=float(10.**rng.uniform(0.5, 1.7)),
                   jitter=float(rng.choice([0., 0.3])), interp_k='log' if log_k else 'lin', extrap_pk='log' if (log_k and i % 3 != 2) else 'lin',
                   extrap_down=int(rng.integers(1, 4)), extrap_up=int(rng.integers(0, 2)), interp_order_k=int(rng.choice([1, 3, 3])),
                   tilt=float(rng.uniform(0.9, 1.05)), knee=float(10.**rng.uniform(-2., -1.3)), wiggle=float(rng.uniform(0., 0.06)))
        if two_d:
            cfg['nz'] = int(rng.choice([1, 6, 12, 30]))
            cfg['zmax'] = float(rng.uniform(1., 4.))
            cfg['interp_order_z'] = int(rng.choice([1, 2, 3, 3, 5])) if cfg['nz'] > 5 else 3
            cfg['growth'] = bool(cfg['nz'] == 1 or i % 4 == 1)
        # The extrapolation range, beyond the table (a range that ends INSIDE the table puts two knots 1e-9 apart at the table's end: the reference's own
        # spline carries ~1e-7 of rounding there): powers of ten, as the reference's defaults (1e-7, 1e2) are, for a third of the configurations, any
        # number for the others.  The reference takes its knots through 10**log10(k): for an end of the range that does not survive that round trip P
        # at the end itself is NaN, and with it every sigma / xi (their FFTLog grid starts and stops there) -- part of what is recorded.
        cfg['extrap_kmin'] = float(10.**(np.floor(np.log10(cfg['kmin'])) - cfg.pop('extrap_down')))
        cfg['extrap_kmax'] = float(10.**(np.ceil(np.log10(cfg['kmax'])) + cfg.pop('extrap_up')))
        if i % 3:
            cfg['extrap_kmin'] *= 0.37 + cfg['wiggle']
            cfg['extrap_kmax'] *= 1.9 + cfg['wiggle']
        configs.append(cfg)
    return configs


def interp_fuzz_table(cfg):
    """(k, z or None, pk table, growth_factor_sq or None) of a configuration: plain numpy, the same for the generator and the tests."""
    rng = np.random.default_rng(int(cfg['nk'] * 1000 + cfg['kmin'] * 1e9) % (2**31))
    logk = np.linspace(np.log10(cfg['kmin']), np.log10(cfg['kmax']), cfg['nk'])
    if cfg['jitter']:
        logk[1:-1] += cfg['jitter'] * (logk[1] - logk[0]) * rng.uniform(-1., 1., cfg['nk'] - 2)
    k = 10.**logk
    pk = 2e4 * (k / cfg['knee'])**cfg['tilt'] / (1. + (k / cfg['knee'])**2)**1.9 * (1. + cfg['wiggle'] * np.sin(70. * k) * np.exp(-(3. * k)**2))
    if not cfg['two_d']:
        return k, None, pk, None
    growth = (lambda zz: 1. / (1. + 0.8 * np.asarray(zz, dtype='f8'))**1.7) if cfg['growth'] else None
    if cfg['nz'] == 1:
        return k, np.array([0.]), pk[:, None], growth
    z = np.sort(np.concatenate([[0.], rng.uniform(0., cfg['zmax'], cfg['nz'] - 2), [cfg['zmax']]]))
    table = pk[:, None] * (1. + 0.1 * np.log10(k / cfg['knee'])[:, None] * z / (1. + z)) / (1. + z)**(0. if cfg['growth'] else 1.5)
    return k, z, table, growth


def interp_fuzz_build(mod, cfg):
    """The interpolator of module ``mod`` (the reference or this package) for a configuration."""
    k, z, pk, growth = interp_fuzz_table(cfg)
    kw = dict(interp_k=cfg['interp_k'], extrap_pk=cfg['extrap_pk'], extrap_kmin=cfg['extrap_kmin'], extrap_kmax=cfg['extrap_kmax'], interp_order_k=cfg['interp_order_k'])
    if not cfg['two_d']:
        return mod.PowerSpectrumInterpolator1D(k, pk, **kw)
    return mod.PowerSpectrumInterpolator2D(k, z, pk, interp_order_z=cfg['interp_order_z'], growth_factor_sq=growth, **kw)


def interp_fuzz_queries(cfg):
    """Wavenumbers (inside the table, in the extrapolation range, outside everything), redshifts (inside and outside), radii."""
    # (the ends of the table and of the extrapolation range themselves among them: the reference goes through 10**log10(k) for its knots, and whether
    # a query AT an end of the range is inside -- a number -- or outside -- NaN -- depends on how that rounds)
    kq = np.concatenate([np.geomspace(cfg['kmin'] * 1.0001, cfg['kmax'] / 1.0001, 23), [cfg['kmin'] * 0.7, cfg['kmax'] * 1.3, cfg['extrap_kmin'] * 0.5,
                         cfg['extrap_kmax'] * 2., 1e-9, 1e4, cfg['kmin'], cfg['kmax'], cfg['extrap_kmin'], cfg['extrap_kmax']]])
    zq = np.array([0., 0.3 * cfg.get('zmax', 1.), 0.77 * cfg.get('zmax', 1.), cfg.get('zmax', 1.), 1.2 * cfg.get('zmax', 1.), -0.1])
    return kq, zq, np.array([2., 8., 30.])


def interp_fuzz_outputs(mod, cfg):
    """What is recorded of a configuration: a dict of arrays (an exception becomes its class name as a string array)."""
    import warnings
    out = {}
    with warnings.catch_warnings(), np.errstate(all='ignore'):
        warnings.simplefilter('ignore')
        interp = interp_fuzz_build(mod, cfg)
        kq, zq, rq = interp_fuzz_queries(cfg)

        def record(name, fn):
            try:
                out[name] = np.asarray(fn(), dtype='f8')
            except Exception as exc:
                out[name] = np.array(type(exc).__name__)

        if not cfg['two_d']:
            record('pk', lambda: interp(kq))
            record('sigma_r', lambda: interp.sigma_r(rq))
            record('sigma8', lambda: interp.sigma8())
            record('sigma_d', lambda: interp.sigma_d())
            record('xi', lambda: interp.to_xi()(np.geomspace(1., 150., 12)))
        else:
            record('pk', lambda: interp(kq, zq))
            record('pk_pairs', lambda: interp(kq[:6], zq, grid=False))
            record('pk_nogrowth', lambda: interp(kq, zq, ignore_growth=True))
            zin = zq[:4] if cfg['nz'] > 1 else zq[:1]
            record('sigma_rz', lambda: interp.sigma_rz(rq, zin))
            record('sigma8_z', lambda: interp.sigma8_z(zin))
            record('sigma_dz', lambda: interp.sigma_dz(zin))
            record('to_1d', lambda: interp.to_1d(z=zin[1 % zin.size])(kq))
            record('xi', lambda: interp.to_xi()(np.geomspace(1., 150., 12), zin))
    return out


def gen_interp_fuzz(cp):
    out = {}
    for i, cfg in enumerate(interp_fuzz_configs()):
        for name, value in interp_fuzz_outputs(cp, cfg).items():
            out['c%d_%s' % (i, name)] = value
    save('interp_fuzz', **out)


# The filters that take options -- hinton2017 (degree, sigma, weight), ehpoly (krange, rescale_krange), kirkby2013 (the two side bands, rescale_sbox),
# the number of wavenumbers of any P(k) filter -- with the options, the cosmology and the fiducial cosmology drawn at random.
FILTER_FUZZ_N = 24


def filter_fuzz_configs(n=FILTER_FUZZ_N, seed=20261007):
    rng = np.random.default_rng(seed)
    configs = []
    for i in range(n):
        cosmo = dict(h=float(rng.uniform(0.6, 0.8)), Omega_m=float(rng.uniform(0.25, 0.38)), Omega_b=float(rng.uniform(0.04, 0.06)), n_s=float(rng.uniform(0.92, 1.)))
        engine = ['hinton2017', 'ehpoly', 'kirkby2013', 'wallish2018', 'savgol', 'peakaverage'][i % 6]
        cfg = dict(engine=engine, cosmo=cosmo, fid_h=float(rng.uniform(0.65, 0.72)), kwargs={})
        if engine == 'hinton2017':
            cfg['kwargs'] = dict(degree=int(rng.choice([9, 12, 13])), sigma=float(rng.uniform(0.3, 1.)), weight=float(rng.uniform(0.5, 0.95)))
        if engine == 'ehpoly':
            cfg['kwargs'] = dict(krange=(float(10.**rng.uniform(-3.3, -2.7)), float(rng.uniform(0.6, 1.5))), rescale_krange=bool(i % 4 < 2))
        if engine == 'kirkby2013':
            left = float(rng.uniform(45., 55.))
            cfg['kwargs'] = dict(srange_left=(left, left + float(rng.uniform(25., 35.))), srange_right=(float(rng.uniform(140., 155.)), float(rng.uniform(180., 200.))),
                                 rescale_sbox=bool(i % 4 < 2))
        if engine in ('wallish2018', 'savgol', 'peakaverage'):
            cfg['nk'] = int(rng.choice([512, 1024, 2048]))
        configs.append(cfg)
    return configs


def filter_fuzz_output(cp, cfg):
    """(coordinates, smooth spectrum / correlation function) of a configuration with the package ``cp`` (the reference or this one), every 8th sample."""
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        cosmo = cp.Cosmology(engine='eisenstein_hu', **cfg['cosmo'])
        fid = cp.Cosmology(engine='eisenstein_hu', h=cfg['fid_h'])
        pk = cosmo.get_fourier().pk_interpolator().to_1d(z=0.)
        if cfg['engine'] == 'kirkby2013':
            xi = pk.clone(extrap_kmin=1e-5).to_xi()      # (the default range of the reference's own xi-side tests)
            f = cp.CorrelationFunctionBAOFilter(xi, engine='kirkby2013', cosmo=cosmo, cosmo_fid=fid, **cfg['kwargs'])
            return np.asarray(f.s)[::8], np.asarray(f.xinow)[::8]
        f = cp.PowerSpectrumBAOFilter(pk, engine=cfg['engine'], cosmo=cosmo, cosmo_fid=fid, **cfg['kwargs'])
        if 'nk' in cfg:      # another number of wavenumbers, and the filter re-used on them (set_k + __call__: bao_filter.py:61-75, 99-107)
            f.set_k(nk=cfg['nk'])
            f(pk)
        return np.asarray(f.k)[::8], np.asarray(f.pknow)[::8]


def filter_fuzz_plateau_reading(cp, cfg):
    """peakaverage: did the reference count the sample in front of the end plateau of its fiducial wiggles as an extremum (bao_filter.py:556-564)?  Its
    fit pins the last two samples of the fitted range to one value; whether the sample before them is a peak for ``find_peaks`` is decided by the last
    bit of its own arithmetic -- recorded, because no other implementation can reproduce that bit (cosmoprimo_amd/bao_filter.py: _wiggle_extrema)."""
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        cosmo = cp.Cosmology(engine='eisenstein_hu', **cfg['cosmo'])
        fid = cp.Cosmology(engine='eisenstein_hu', h=cfg['fid_h'])
        f = cp.PowerSpectrumBAOFilter(cosmo.get_fourier().pk_interpolator().to_1d(z=0.), engine='peakaverage', cosmo=cosmo, cosmo_fid=fid, **cfg['kwargs'])
        # (a filter re-used on another number of wavenumbers keeps the extrema of the grid it was built on: `_prepare` runs once, bao_filter.py:61-75)
        index = np.flatnonzero((f.k >= 1e-3) & (f.k <= 1.))
        before_plateau = f.k[index[-1] - 1]
        return bool(any(f.k_peaks[i][f.pad_peaks[i][0] + f.pad_peaks[i][1] - 1] == before_plateau for i in (0, 1)))


def gen_filter_fuzz(cp):
    out = {}
    for i, cfg in enumerate(filter_fuzz_configs()):
        out['c%d_x' % i], out['c%d_smooth' % i] = filter_fuzz_output(cp, cfg)
        if cfg['engine'] == 'peakaverage':
            out['c%d_plateau_extremum' % i] = np.array(filter_fuzz_plateau_reading(cp, cfg))
    save('filter_fuzz', **out)


# Parameter conventions drawn at random: the same cosmology can be given as h or H0, Omega_x or omega_x, Omega_m or Omega_cdm, m_ncdm (a list, or a
# sum with a hierarchy) or Omega_ncdm, N_eff or N_ur, T_cmb or Omega_g, A_s or ln10^{10}A_s or logA or sigma8, w0 / wa ... (reference cosmology.py:
# compile_params, :1049-1260): what the compiled cosmology holds for a fixed list of names.
PARAMS_FUZZ_N = 40
PARAMS_FUZZ_NAMES = ['h', 'H0', 'Omega_b', 'omega_b', 'Omega_cdm', 'omega_cdm', 'Omega_m', 'omega_m', 'Omega_k', 'Omega_g', 'omega_g', 'T_cmb', 'Omega_ur', 'N_ur',
                     'N_eff', 'Omega_ncdm_tot', 'omega_ncdm_tot', 'Omega_pncdm_tot', 'm_ncdm_tot', 'N_ncdm', 'Omega_de', 'Omega_Lambda', 'Omega_fld', 'w0_fld', 'wa_fld',
                     'n_s', 'alpha_s', 'k_pivot', 'Omega_r']


def params_fuzz_configs(n=PARAMS_FUZZ_N, seed=20261008):
    rng = np.random.default_rng(seed)
    configs = []
    for i in range(n):
        h = float(rng.uniform(0.55, 0.85))
        par = {}
        par.update({'h': h} if i % 2 else {'H0': 100. * h})
        ob = float(rng.uniform(0.04, 0.06))
        par.update({'Omega_b': ob} if i % 3 else {'omega_b': ob * h**2})
        kind = i % 4
        if kind == 0:
            par['Omega_cdm'] = float(rng.uniform(0.2, 0.35))
        elif kind == 1:
            par['omega_cdm'] = float(rng.uniform(0.2, 0.35)) * h**2
        elif kind == 2:
            par['Omega_m'] = float(rng.uniform(0.25, 0.4))
        else:
            par['omega_m'] = float(rng.uniform(0.25, 0.4)) * h**2
        nu = (i // 4) % 5
        if nu == 1:
            par['m_ncdm'] = [0.06]
        elif nu == 2:
            par['m_ncdm'] = [float(rng.uniform(0.01, 0.1)), float(rng.uniform(0.05, 0.3))]
        elif nu == 3:
            par['m_ncdm'] = float(rng.uniform(0.07, 0.4))
            par['neutrino_hierarchy'] = ['normal', 'inverted', 'degenerate'][i % 3]
        elif nu == 4:
            par['m_ncdm'] = [0.06]
            par['T_ncdm_over_cmb'] = [float(rng.uniform(0.68, 0.73))]
        if i % 5 == 1:
            par['N_eff'] = float(rng.uniform(2.8, 4.))
        elif i % 5 == 2 and nu == 0:
            par['N_ur'] = float(rng.uniform(2.5, 3.5))
        if i % 7 == 3:
            par['T_cmb'] = float(rng.uniform(2.6, 2.8))
        elif i % 7 == 5:
            par['Omega_g'] = float(rng.uniform(4.5e-5, 6e-5))
        if i % 3 == 1:
            par['Omega_k'] = float(rng.uniform(-0.1, 0.1))
        if i % 6 in (2, 3):
            par['w0_fld'] = float(rng.uniform(-1.3, -0.7))
        if i % 6 == 3:
            par['wa_fld'] = float(rng.uniform(-0.5, 0.3))
        amp = i % 4
        if amp == 0:
            par['A_s'] = float(rng.uniform(1.8e-9, 2.4e-9))
        elif amp == 1:
            par['ln10^{10}A_s'] = float(rng.uniform(2.9, 3.2))
        elif amp == 2:
            par['logA'] = float(rng.uniform(2.9, 3.2))
        else:
            par['sigma8'] = float(rng.uniform(0.7, 0.9))
        par['n_s'] = float(rng.uniform(0.92, 1.))
        if i % 8 == 5:
            par['alpha_s'] = float(rng.uniform(-0.02, 0.02))
        configs.append(par)
    return configs


def params_fuzz_output(cp, par):
    """The compiled parameters PARAMS_FUZZ_NAMES of ``Cosmology(**par)`` and what the Eisenstein-Hu engine makes of its amplitude (A_s, sigma8, rs_drag)."""
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        cosmo = cp.Cosmology(engine='eisenstein_hu', **par)
        values = [float(np.sum(cosmo[name])) if name == 'm_ncdm_tot' or np.ndim(cosmo[name]) else float(cosmo[name]) for name in PARAMS_FUZZ_NAMES]
        values += [float(cosmo.get_primordial().A_s), float(cosmo.get_fourier().sigma8_m), float(cosmo.get_thermodynamics().rs_drag)]
        masses = np.asarray(cosmo['m_ncdm'], dtype='f8').ravel()
        return np.array(values), np.concatenate([masses, np.full(3 - masses.size, np.nan)])


def gen_params_fuzz(cp):
    values, masses = zip(*[params_fuzz_output(cp, par) for par in params_fuzz_configs()])
    save('params_fuzz', values=np.stack(values), m_ncdm=np.stack(masses))


# Tabulated xi(s) / xi(s, z) interpolators with options drawn at random (separations geometric or jittered, interpolation in s or log s, spline degrees,
# growth factor or table in z): values inside and outside, pairs, to_pk and the sigma functions that go through it.
XI_FUZZ_N = 24


def xi_fuzz_configs(n=XI_FUZZ_N, seed=20261009):
    rng = np.random.default_rng(seed)
    configs = []
    for i in range(n):
        cfg = dict(two_d=bool(i % 2), ns=int(rng.choice([60, 200, 500])), smin=float(10.**int(rng.integers(-3, 0))), smax=float(10.**int(rng.integers(3, 5))),
                   jitter=float(rng.choice([0., 0.3])), interp_s='log' if i % 4 != 3 else 'lin', interp_order_s=int(rng.choice([1, 3, 3])),
                   slope=float(rng.uniform(1.5, 2.1)), bump=float(rng.uniform(0., 0.01)))
        if cfg['two_d']:
            cfg['nz'] = int(rng.choice([1, 6, 12]))
            cfg['zmax'] = float(rng.uniform(1., 3.))
            cfg['interp_order_z'] = int(rng.choice([1, 3, 3])) if cfg['nz'] > 3 else 3
            cfg['growth'] = bool(cfg['nz'] == 1 or i % 4 == 1)
        if i % 3:      # ends of the table that are not powers of ten (see interp_fuzz_configs)
            cfg['smin'] *= 0.37 + 10. * cfg['bump']
            cfg['smax'] *= 1.9 + 10. * cfg['bump']
        configs.append(cfg)
    return configs


def xi_fuzz_build(mod, cfg):
    rng = np.random.default_rng(int(cfg['ns'] * 1000 + cfg['smin'] * 1e7) % (2**31))
    logs = np.linspace(np.log10(cfg['smin']), np.log10(cfg['smax']), cfg['ns'])
    if cfg['jitter']:
        logs[1:-1] += cfg['jitter'] * (logs[1] - logs[0]) * rng.uniform(-1., 1., cfg['ns'] - 2)
    s = 10.**logs
    xi = (s / 5.)**(-cfg['slope']) * np.exp(-(s / 300.)**2) + cfg['bump'] * np.exp(-(s - 100.)**2 / 200.) - 2e-4 * np.exp(-(s / 600.)**2)
    kw = dict(interp_s=cfg['interp_s'], interp_order_s=cfg['interp_order_s'])
    if not cfg['two_d']:
        return mod.CorrelationFunctionInterpolator1D(s, xi, **kw), s
    growth = (lambda zz: 1. / (1. + 0.8 * np.asarray(zz, dtype='f8'))**1.7) if cfg['growth'] else None
    if cfg['nz'] == 1:
        return mod.CorrelationFunctionInterpolator2D(s, np.array([0.]), xi[:, None], interp_order_z=3, growth_factor_sq=growth, **kw), s      # (the reference's default None does not construct)
    z = np.sort(np.concatenate([[0.], rng.uniform(0., cfg['zmax'], cfg['nz'] - 2), [cfg['zmax']]]))
    table = xi[:, None] * (1. + 0.05 * np.log10(s / 10.)[:, None] * z / (1. + z)) / (1. + z)**(0. if cfg['growth'] else 1.5)
    return mod.CorrelationFunctionInterpolator2D(s, z, table, interp_order_z=cfg['interp_order_z'], growth_factor_sq=growth, **kw), s


def xi_fuzz_outputs(mod, cfg):
    import warnings
    out = {}
    with warnings.catch_warnings(), np.errstate(all='ignore'):
        warnings.simplefilter('ignore')
        interp, s = xi_fuzz_build(mod, cfg)
        sq = np.concatenate([np.geomspace(cfg['smin'] * 1.0001, cfg['smax'] / 1.0001, 19), [cfg['smin'] * 0.5, cfg['smax'] * 2., s[0], s[-1]]])
        zq = np.array([0., 0.3 * cfg.get('zmax', 1.), 0.77 * cfg.get('zmax', 1.), 1.2 * cfg.get('zmax', 1.), -0.1])
        kq = np.geomspace(2e-3, 3., 15)

        def record(name, fn):
            try:
                out[name] = np.asarray(fn(), dtype='f8')
            except Exception as exc:
                out[name] = np.array(type(exc).__name__)

        if not cfg['two_d']:
            record('xi', lambda: interp(sq))
            record('pk', lambda: interp.to_pk()(kq))
            record('sigma8', lambda: interp.sigma8())
        else:
            zin = zq[:3] if cfg['nz'] > 1 else zq[:1]
            record('xi', lambda: interp(sq, zq))
            record('xi_pairs', lambda: interp(sq[:5], zq, grid=False))
            record('xi_nogrowth', lambda: interp(sq, zq, ignore_growth=True))
            record('to_1d', lambda: interp.to_1d(z=zin[1 % zin.size])(sq))
            record('pk', lambda: interp.to_pk()(kq, zin))
            record('sigma8_z', lambda: interp.sigma8_z(zin))
    return out


def gen_xi_fuzz(cp):
    out = {}
    for i, cfg in enumerate(xi_fuzz_configs()):
        for name, value in xi_fuzz_outputs(cp, cfg).items():
            out['c%d_%s' % (i, name)] = value
    save('xi_fuzz', **out)


def main():
    os.makedirs(OUT, exist_ok=True)
    cp = import_reference()
    which = sys.argv[1:] or ['fftlog']
    if 'fftlog' in which:
        pks = gen_pk_eh(cp)
        gen_fftlog_tables(cp)
        gen_loggamma()
        gen_fftlog_transforms(cp, pks)
    if 'background' in which:
        gen_background(cp)
    if 'power' in which:
        gen_power(cp)
    if 'sigma' in which:
        gen_sigma(cp)
    if 'sigma_quad' in which:
        gen_sigma_quad(cp)
    if 'sigma_api' in which:
        gen_sigma_api(cp)
    if 'api_signatures' in which:
        gen_api_signatures(cp)
    if 'bao' in which:
        gen_bao(cp)
    if 'xi' in which:
        gen_xi(cp)
    if 'bao2' in which:
        gen_bao2(cp)
    if 'bspline' in which:
        gen_bspline(cp)
    if 'densities' in which:
        gen_densities(cp)
    if 'ncdm' in which:
        gen_ncdm(cp)
    if 'variants' in which:
        gen_variants(cp)
    if 'power_ncdm' in which:
        gen_power_ncdm(cp)
    if 'bao_batch' in which:
        gen_bao_batch(cp)
    if 'fuzz' in which:
        gen_fuzz(cp)
    if 'fftlog_fuzz' in which:
        gen_fftlog_fuzz(cp)
    if 'fftlog_large' in which:
        gen_fftlog_large(cp)
    if 'interp_fuzz' in which:
        gen_interp_fuzz(cp)
    if 'filter_fuzz' in which:
        gen_filter_fuzz(cp)
    if 'params_fuzz' in which:
        gen_params_fuzz(cp)
    if 'xi_fuzz' in which:
        gen_xi_fuzz(cp)
    if 'calculator' in which:
        gen_calculator(cp)
    if 'cosmology_api' in which:
        gen_cosmology_api(cp)
    if 'abacus' in which:
        gen_abacus_table(cp)
    if 'desi_table' in which:
        gen_desi_table()
    if 'api_flows' in which:
        gen_api_flows(cp)


if __name__ == '__main__':
    main()
