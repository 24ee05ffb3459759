"""Pin the oracle restatement of the eisenstein_hu_nowiggle_variants engine (oracle/power.py: variants_*)
against golden vectors from the reference (tests/golden/variants.npz)."""
import numpy as np
import pytest

from oracle import background as ob
from oracle import power as op
from oracle.gen_golden import VARIANTS_PARAMS

NAMES = ['omega_b', 'omega_m', 'frac_b', 'frac_cdm', 'frac_cb', 'frac_ncdm', 'theta_cmb', 'z_eq', 'k_eq', 'z_drag', 'rs_drag', 'p_c', 'p_cb', 'gamma_ncdm', 'beta_c']


def background_params(par):
    par = {k: v for k, v in par.items() if k in ('m_ncdm', 'Omega_m', 'h', 'Omega_b', 'T_cmb', 'Omega_cdm')}
    if 'm_ncdm' in par:
        return ob.derived_ncdm(par.pop('m_ncdm'), **par)
    p = {k: v for k, v in ob.derived(**par).items()}
    p['T_cmb'] = par.get('T_cmb', ob.TCMB)
    return p


@pytest.mark.parametrize('ic', range(len(VARIANTS_PARAMS)))
def test_variants(golden, ic):
    g = golden('variants')
    k, z = g['k'], g['z']
    p = background_params(VARIANTS_PARAMS[ic])
    s = op.variants_scalars(p)
    for name in NAMES:
        np.testing.assert_allclose(s[name], g['c%d_%s' % (ic, name)], rtol=1e-13, err_msg=name)
    growth_k0 = g['c%d_growth_k0' % ic]
    for of in ['delta_m', 'delta_cb']:
        np.testing.assert_allclose(op.variants_transfer_kz(k, z, p, s, growth_k0, of=of), g['c%d_transfer_%s' % (ic, of)], rtol=1e-12)
