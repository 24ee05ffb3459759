"""Pin the massive-neutrino oracle (oracle/background.py: ncdm_*) against golden vectors from the reference."""
import numpy as np
import pytest

from oracle import background as ob
from oracle.gen_golden import NCDM_PARAMS


def oracle_params(par):
    par = dict(par)
    m = par.pop('m_ncdm')
    t = par.pop('T_ncdm_over_cmb', None)
    return ob.derived_ncdm(m, T_ncdm_over_cmb=None if t is None else np.asarray(t, dtype='f8'), **par)


@pytest.mark.parametrize('ic', range(len(NCDM_PARAMS)))
def test_ncdm(golden, ic):
    g = golden('ncdm')
    z = g['z']
    p = oracle_params(NCDM_PARAMS[ic])
    pre = 'c%d_' % ic
    np.testing.assert_allclose(p['N_ur'], g[pre + 'par_N_ur'], rtol=1e-14)
    np.testing.assert_allclose(p['Omega_ncdm'], g[pre + 'par_Omega_ncdm'], rtol=1e-13)
    np.testing.assert_allclose(sum(p['Omega_pncdm']), g[pre + 'par_Omega_pncdm_tot'], rtol=1e-13)
    np.testing.assert_allclose(p['Omega_de'], g[pre + 'par_Omega_de'], rtol=1e-13)
    np.testing.assert_allclose(p['Omega_cdm'], g[pre + 'par_Omega_cdm'], rtol=1e-13)
    np.testing.assert_allclose(p['T_ncdm'], g[pre + 'par_T_ncdm'], rtol=1e-15)
    np.testing.assert_allclose(ob.ncdm_interp(p, z, 'rho'), g[pre + 'rho_ncdm'], rtol=1e-12)
    np.testing.assert_allclose(ob.ncdm_interp(p, z, 'p'), g[pre + 'p_ncdm'], rtol=1e-12)
    np.testing.assert_allclose(ob.efunc_ncdm(z, p), g[pre + 'efunc'], rtol=1e-13)
    np.testing.assert_allclose(ob.comoving_radial_distance_ncdm(z, p), g[pre + 'comoving_radial_distance'], rtol=1e-12, atol=1e-300)


def test_ncdm_tables(golden):
    g = golden('ncdm')
    zc, rho, _ = ob.ncdm_tables(oracle_params(NCDM_PARAMS[-1]))
    np.testing.assert_allclose(zc, g['ncdm_knots'], rtol=1e-15)
    np.testing.assert_allclose(rho.T, g['rho_ncdm_table'], rtol=1e-13)
