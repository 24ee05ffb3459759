"""Reference values of sampled units of BASELINE.json's configs 3, 3B, 4 and 5 from the oracle (TEST / BENCH INFRASTRUCTURE: the checker
behind tests/test_full_size_gpu.py and behind the untimed, after-the-clock ``parity_spot_check`` of every secondary config on bench.py's
line; the product never imports it).  Each function restates the route the reference takes for ONE unit (file:line in the oracle modules it
calls) and costs at most a few tenths of a second."""
import os

import numpy as np
from scipy.interpolate import CubicSpline

from . import background as ob, bao as obao, interp as oi, power as op, sigma as osg

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests', 'golden')

# tolerances of SURVEY.md 8(d), pointwise relative
TOLERANCES = {'config3': 1e-9, 'config3b': 1e-9, 'config4': 1e-9, 'config5': 1e-10}


def eh_pk(par, engine='eisenstein_hu'):
    """Normalised P(k, z=0) callable of an analytic engine for reference-style parameters (oracle power + sigma pipeline), and its rs_drag [Mpc/h]."""
    par = dict(par)
    h, Ob = par.get('h', 0.7), par.get('Omega_b', 0.05)
    Ocdm = par['Omega_m'] - Ob if 'Omega_m' in par else 0.25
    s8, ns = par.get('sigma8', 0.8), par.get('n_s', 0.96)
    p = ob.derived(h=h, Omega_cdm=Ocdm, Omega_b=Ob)
    D0 = op.growth_factor(0., p, znorm=0.)

    def raw(k):
        return op.pk_z0(k, engine, h=h, Omega_cdm=Ocdm, Omega_b=Ob, sigma8=s8, n_s=ns) * D0**2

    rs = s8 / np.sqrt(osg.sigma_r2(8., raw))
    return (lambda k: np.where((k >= 1e-7) & (k <= 1e2), raw(k) * rs**2, np.nan)), op.eh_scalars(h, Ocdm, Ob)['rs_drag'] * h


def pad_log_natural_eval(kk, pp, ke):
    """PowerSpectrumInterpolator1D(kk, pp)(ke): _pad_log + natural cubic spline in log10-log10 (interpolator.py:42-87, 419-451; jax.py:172)."""
    logk, logp = np.log10(kk), np.log10(pp)
    lmin, lmax = np.log10(np.minimum(1e-7, kk[0] * (1 - 1e-9))), np.log10(np.maximum(1e2, kk[-1] * (1 + 1e-9)))
    sl = (logp[-1] - logp[-2]) / (logk[-1] - logk[-2])
    hk = np.array([logk[-1] * 0.1 + lmax * 0.9, lmax])
    hp = np.array([logp[-1] + sl * (hk[0] - logk[-1]), logp[-1] + sl * (hk[1] - logk[-1])])
    sl = (logp[1] - logp[0]) / (logk[1] - logk[0])
    lk = np.array([lmin, logk[0] * 0.1 + lmin * 0.9])
    lp = np.array([logp[0] + sl * (lk[0] - logk[0]), logp[0] + sl * (lk[1] - logk[0])])
    x, y = np.concatenate([lk, logk, hk]), np.concatenate([lp, logp, hp])
    return 10**CubicSpline(x, y, axis=0, bc_type='natural')(np.log10(ke))


def config3_sigma_rz(par, r, z, sigma8=0.8):
    """sigma(r, z), (nr, nz), of ONE EH98 cosmology ``par`` = dict(Omega_m, Omega_b, h, n_s) normalised to ``sigma8`` at z = 0: the reference's
    path for a single cosmology (P(k) x growth^2 per redshift, one TophatVariance FFTLog per column, natural spline to r; interpolator.py:846-875)."""
    Om, Ob, h, ns = (float(par[name]) for name in ('Omega_m', 'Omega_b', 'h', 'n_s'))
    g2 = op.growth_factor(np.asarray(z, dtype='f8'), ob.derived(h=h, Omega_b=Ob, Omega_m=Om), znorm=0.)**2
    pk0 = lambda k: op.pk_z0(k, 'eisenstein_hu', h=h, Omega_cdm=Om - Ob, Omega_b=Ob, n_s=ns)        # noqa: E731
    g0 = op.growth_factor(np.zeros(1), ob.derived(h=h, Omega_b=Ob, Omega_m=Om), znorm=0.)[0]**2
    norm = sigma8**2 / (float(osg.sigma_r2(np.array([8.]), pk0)[0]) * g0)      # sigma8 is set at z = 0, growth factor (not 1 there) included
    return (norm * osg.sigma_r2(np.asarray(r, dtype='f8'), lambda k: pk0(k)[:, None] * g2[None, :]))**0.5


def config3b_sigma_rz(k, z, table, r, zq):
    """sigma(r, z), (nr, nz), of ONE tabulated P(k, z): RectBivariateSpline of log10 P on (log10 k, z) with the log-log padding, P(k, z) at the
    redshifts, one TophatVariance FFTLog per redshift, natural spline to r (interpolator.py:609-700, 846-875)."""
    pk2d = oi.pk_interp_2d(k, z, table)
    return np.sqrt(osg.sigma_r2(np.asarray(r, dtype='f8'), lambda kk: pk2d(kk, np.asarray(zq, dtype='f8'), grid=True)))


_brieden_prep = {}


def brieden2022_prepared():
    """The ``_prepare`` products of brieden2022 for the default fiducial (bao_filter.py:461-491), with the envelope-knot lists the reference itself
    produced (tests/golden/bao.npz: one of its knots is decided by rounding, tests/test_oracle_bao.py).  Returns (prep, rs_drag_fid)."""
    if not _brieden_prep:
        pk_fid, rs_fid = eh_pk({})
        pknow_fid, _ = eh_pk({}, 'eisenstein_hu_nowiggle')
        prep = obao.brieden2022_prepare(pk_fid, pknow_fid)
        gold = np.load(os.path.join(GOLDEN_DIR, 'bao.npz'))
        prep['peaks'] = [gold['brieden_peaks_high'], gold['brieden_peaks_low']]
        prep['ratio_now_fid'] = obao._interp_envelopes(*prep['peaks'], prep['k_fid'], prep['ratio_fid'])
        _brieden_prep['value'] = (prep, rs_fid)
    return _brieden_prep['value']


def config4_pknow(par, rsigma8, engine):
    """The smooth spectrum (1024 wavenumbers) ``engine`` in ('wallish2018', 'brieden2022') returns for ONE EH98 cosmology ``par`` whose spectrum
    carries the normalisation factor ``rsigma8`` the package found (the filters work on the growth-less P(k) of a 2-D interpolator, reference
    bao_filter.py:363, 493: ignore_growth=True; bao_filter.py:361-431, 461-509)."""
    p = {name: float(par[name]) for name in ('Omega_m', 'Omega_b', 'h', 'n_s')}

    def pk(k):
        return op.pk_z0(k, 'eisenstein_hu', h=p['h'], Omega_cdm=p['Omega_m'] - p['Omega_b'], Omega_b=p['Omega_b'], n_s=p['n_s'], rsigma8=float(rsigma8))

    if engine == 'wallish2018':
        return obao.wallish2018(lambda k: pk(k)[:, None])[:, 0]
    prep, rs_fid = brieden2022_prepared()
    _, rs = eh_pk(p)
    pknow_c, _ = eh_pk(p, 'eisenstein_hu_nowiggle')
    return obao.brieden2022_compute(prep, lambda k: pk(k)[:, None], pknow_c, rs / rs_fid, lambda kk, pp, ke: pad_log_natural_eval(kk, pp[:, 0], ke))[:, 0]


def config5_distances(om, w0, wa, zz):
    """comoving_radial_distance [Mpc/h] of the samples (one fresh cosmology each; cosmology.py:2027-2042 through oracle/background.py)."""
    return np.array([ob.comoving_radial_distance(np.array([z]), ob.derived(Omega_m=a, w0_fld=b, wa_fld=c))[0] for a, b, c, z in zip(om, w0, wa, zz)])


def max_relative_error(got, ref):
    got, ref = np.asarray(got, dtype='f8'), np.asarray(ref, dtype='f8')
    return float(np.max(np.abs(got - ref) / np.abs(ref)))
