"""The oracle on cosmologies drawn from wide priors (curvature, (w0, wa), one to three massive species, N_eff, T_cmb, sigma8 or A_s) against the
reference's own outputs for them (tests/golden/fuzz.npz, `python -m oracle.gen_golden fuzz`): analytic engines and background."""
import numpy as np
import pytest

from oracle import background as ob, power as op
from oracle.gen_golden import FUZZ_N

ENGINES = ['eisenstein_hu', 'eisenstein_hu_nowiggle', 'bbks']


def oracle_params(g, i):
    par = {name: float(g['par_' + name][i]) for name in ['h', 'Omega_cdm', 'Omega_b', 'Omega_k', 'T_cmb', 'N_ur', 'w0_fld', 'wa_fld']}
    m = g['par_m_ncdm'][i]
    keep = np.isfinite(m)
    return ob.derived_ncdm(m[keep], T_ncdm_over_cmb=g['par_T_ncdm_over_cmb'][i][keep], **par)


@pytest.mark.parametrize('i', range(FUZZ_N))
def test_background_of_random_cosmologies(golden, i):
    g = golden('fuzz')
    p = oracle_params(g, i)
    zb = g['zb']
    np.testing.assert_allclose(p['Omega_de'], g['par_Omega_de'][i], rtol=1e-12)
    np.testing.assert_allclose(ob.efunc_ncdm(zb, p), g['efunc'][i], rtol=1e-12)
    np.testing.assert_allclose(ob.comoving_radial_distance_ncdm(zb, p), g['comoving_radial_distance'][i], rtol=1e-10)


@pytest.mark.parametrize('engine', ENGINES)
def test_engines_on_random_cosmologies(golden, engine):
    g = golden('fuzz')
    k, z = g['k'], g['z']
    for i in range(FUZZ_N):
        p = oracle_params(g, i)
        np.testing.assert_allclose(op.growth_factor_ncdm(z, p), g[engine + '_growth_factor'][i], rtol=1e-11, err_msg=str(i))
        np.testing.assert_allclose(op.growth_rate_ncdm(z, p), g[engine + '_growth_rate'][i], rtol=1e-11, err_msg=str(i))
        prim = {name: float(g['par_' + name][i]) for name in ['n_s', 'alpha_s', 'beta_s', 'k_pivot']}
        _, pk0 = op.pk_z0_ncdm(k, p, engine=engine, A_s=float(g[engine + '_A_s_fid'][i]), rsigma8=float(g[engine + '_rsigma8'][i]), **prim)
        g2 = op.growth_factor_ncdm(z, p, znorm=0.)**2
        np.testing.assert_allclose(pk0[:, None] * g2, g[engine + '_pkz'][i], rtol=1e-10, err_msg=str(i))
        if engine != 'bbks':
            s = op.eh_scalars(p['h'], p['Omega_cdm'], p['Omega_b'], p['T_cmb'])
            np.testing.assert_allclose(s['rs_drag'] * p['h'], g[engine + '_rs_drag'][i], rtol=1e-12)
            np.testing.assert_allclose(s['z_drag'], g[engine + '_z_drag'][i], rtol=1e-12)
