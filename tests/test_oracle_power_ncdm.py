"""Pin the oracle of the analytic engines on cosmologies WITH massive neutrinos (oracle/power.py: *_ncdm; SURVEY.md 8(a) a26-a31 with a23) against
golden vectors from the reference (tests/golden/power_ncdm.npz, `python -m oracle.gen_golden power_ncdm`): the reference computes for any N_ncdm
(eisenstein_hu.py:21-33, its warnings are commented out); fiducial.DESI() and the CosmoSIS default mnu = 0.06 are such cosmologies."""
import numpy as np
import pytest

from oracle import background as ob, power as op, sigma as osig
from oracle.gen_golden import POWER_NCDM_CASES

ENGINES = ['eisenstein_hu', 'eisenstein_hu_nowiggle', 'bbks']


def oracle_params(g, pre):
    """oracle.background.derived_ncdm of a golden case from its compiled parameters (what the kernels are handed as well)."""
    par = {name: float(g[pre + 'par_' + name]) for name in ['h', 'Omega_cdm', 'Omega_b', 'Omega_k', 'T_cmb', 'N_ur', 'w0_fld', 'wa_fld']}
    return ob.derived_ncdm(g[pre + 'par_m_ncdm'], T_ncdm_over_cmb=g[pre + 'par_T_ncdm_over_cmb'], **par)


def primordial(g, pre):
    return {name: float(g[pre + 'par_' + name]) for name in ['n_s', 'alpha_s', 'beta_s', 'k_pivot']}


@pytest.mark.parametrize('engine', ENGINES)
@pytest.mark.parametrize('ic', range(len(POWER_NCDM_CASES)))
def test_analytic_engines_with_massive_neutrinos(golden, engine, ic):
    g = golden('power_ncdm')
    k, z = g['k'], g['z']
    pre = '%s_c%d_' % (engine, ic)
    p = oracle_params(g, pre)
    np.testing.assert_allclose(p['Omega_de'], g[pre + 'par_Omega_de'], rtol=1e-13)
    np.testing.assert_allclose(p['Omega_b'] + p['Omega_cdm'] + sum(p['Omega_ncdm']) - sum(p['Omega_pncdm']), g[pre + 'par_Omega_m'], rtol=1e-13)
    np.testing.assert_allclose(op.growth_factor_ncdm(z, p), g[pre + 'growth_factor'], rtol=1e-12)
    np.testing.assert_allclose(op.growth_factor_ncdm(z, p, znorm=0.), g[pre + 'growth_factor_znorm0'], rtol=1e-12)
    np.testing.assert_allclose(op.growth_rate_ncdm(z, p), g[pre + 'growth_rate'], rtol=1e-12)
    rs = float(g[pre + 'rsigma8'])
    tr, pk0 = op.pk_z0_ncdm(k, p, engine=engine, A_s=float(g[pre + 'A_s_fid']), rsigma8=rs, **primordial(g, pre))
    np.testing.assert_allclose(tr, g[pre + 'transfer'], rtol=1e-12)
    np.testing.assert_allclose(float(g[pre + 'A_s_fid']) * rs**2, g[pre + 'A_s'], rtol=1e-13)
    g2 = op.growth_factor_ncdm(z, p, znorm=0.)**2
    f = op.growth_rate_ncdm(z, p)
    np.testing.assert_allclose(pk0[:, None] * g2, g[pre + 'pkz'], rtol=1e-11)
    np.testing.assert_allclose(pk0[:, None] * g2 * f**2, g[pre + 'pkz_theta'], rtol=1e-11)
    np.testing.assert_allclose(pk0[:, None] * g2 * f, g[pre + 'pkz_delta_theta'], rtol=1e-11)
    if engine != 'bbks':      # the fits know nothing of the species: omega_cdm + omega_b (eisenstein_hu.py:37-38)
        s = op.eh_scalars(p['h'], p['Omega_cdm'], p['Omega_b'], p['T_cmb'])
        for name in ['z_eq', 'k_eq', 'z_drag', 'rs_drag'] + (['k_silk', 'alpha_c', 'beta_c', 'alpha_b', 'beta_node', 'beta_b'] if engine == 'eisenstein_hu' else ['alpha_gamma']):
            np.testing.assert_allclose(s[name], g[pre + name], rtol=1e-13, err_msg=name)
        np.testing.assert_allclose(s['rs_drag'] * p['h'], g[pre + 'rs_drag_h'], rtol=1e-13)
    else:
        nu_m = sum(p['Omega_ncdm']) - sum(p['Omega_pncdm'])
        np.testing.assert_allclose(op.bbks_gamma(p['h'], p['Omega_cdm'], p['Omega_b'], nu_m), g[pre + 'gamma'], rtol=1e-13)


@pytest.mark.parametrize('engine', ENGINES)
def test_sigma8_with_massive_neutrinos(golden, engine):
    """sigma8_m, sigma8(z), sigma(r, z) and the rescaling factor of the normalisation (eisenstein_hu.py:94-103, 331-342) for the cases with a sigma8 target."""
    g = golden('power_ncdm')
    z = g['z']
    for ic in (0, 1, 2):
        pre = '%s_c%d_' % (engine, ic)
        p = oracle_params(g, pre)
        A_fid, pm = float(g[pre + 'A_s_fid']), primordial(g, pre)
        g0 = op.growth_factor_ncdm(np.zeros(()), p, znorm=0.)**2

        def pk(kk, rsigma8=1.):
            return op.pk_z0_ncdm(kk, p, engine=engine, A_s=A_fid, rsigma8=rsigma8, **pm)[1]

        s8_fid = np.sqrt(osig.sigma_r2(8., lambda kk: pk(kk) * g0))
        target = float(g[pre + 'sigma8_m'])
        rs = target / s8_fid
        np.testing.assert_allclose(rs, g[pre + 'rsigma8'], rtol=1e-10)
        r = np.array([2., 8., 30.])
        sig0 = np.sqrt(osig.sigma_r2(r, lambda kk: pk(kk, rsigma8=float(g[pre + 'rsigma8']))))
        growth = op.growth_factor_ncdm(z, p, znorm=0.)
        np.testing.assert_allclose(sig0[:, None] * growth, g[pre + 'sigma_rz'], rtol=1e-10)
        np.testing.assert_allclose(sig0[1] * growth, g[pre + 'sigma8_z'], rtol=1e-10)
