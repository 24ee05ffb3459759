"""Pin the analytic P(k) and sigma(r) oracles against golden vectors from the reference (G7, G4)."""
import numpy as np
import pytest

from oracle import background as ob
from oracle import power as op
from oracle import sigma as osg

ENGINES = ['eisenstein_hu', 'eisenstein_hu_nowiggle', 'bbks']


def cosmo_of(g):
    return dict(h=g['h'], Omega_cdm=g['Omega_m'] - g['Omega_b'], Omega_b=g['Omega_b'])


@pytest.mark.parametrize('eng', ENGINES)
def test_power(golden, eng):
    g = golden('power')
    c = cosmo_of(g)
    k, z = g['k'], g['z']
    s = op.eh_scalars(**c)
    if eng != 'bbks':
        for name in ['z_eq', 'k_eq', 'z_drag', 'r_drag', 'r_eq', 'rs_drag'] + (['k_silk', 'alpha_c', 'beta_c', 'alpha_b', 'beta_node', 'beta_b'] if eng == 'eisenstein_hu' else ['alpha_gamma']):
            np.testing.assert_allclose(s[name], g['%s_%s' % (eng, name)], rtol=1e-15)
        np.testing.assert_allclose(s['rs_drag'] * g['h'], g[eng + '_rs_drag_h'], rtol=1e-15)
        tr = op.transfer_eh(k, g['h'], s) if eng == 'eisenstein_hu' else op.transfer_nowiggle(k, g['h'], s)
    else:
        gam = op.bbks_gamma(**c)
        np.testing.assert_allclose(gam, g['bbks_gamma'], rtol=1e-15)
        tr = op.transfer_bbks(k, g['h'], gam)
    np.testing.assert_allclose(tr, g[eng + '_transfer'], rtol=1e-14)
    A = g[eng + '_A_s']
    np.testing.assert_allclose(op.A_s_fid(g['sigma8']) * g[eng + '_rsigma8']**2, A, rtol=1e-15)
    prim = op.primordial_pk(k, g['h'], A, g['n_s'], g['alpha_s'])
    np.testing.assert_allclose(prim, g[eng + '_pk_prim'], rtol=1e-14)
    p = ob.derived(h=g['h'], Omega_m=g['Omega_m'], Omega_b=g['Omega_b'], w0_fld=g['w0_fld'], wa_fld=g['wa_fld'])
    D = op.growth_factor(z[None, :], p, znorm=0.)
    np.testing.assert_allclose(D, g[eng + '_growth_factor_znorm0'], rtol=1e-14)
    np.testing.assert_allclose(op.growth_factor(z[None, :], p), g[eng + '_growth_factor'], rtol=1e-14)
    np.testing.assert_allclose(op.growth_rate(z[None, :], p), g[eng + '_growth_rate'], rtol=1e-14)
    pk0 = op.pk_z0(k, eng, sigma8=g['sigma8'], n_s=g['n_s'], alpha_s=g['alpha_s'], rsigma8=g[eng + '_rsigma8'], **c)
    pkz = pk0[:, :, None] * (D**2)[:, None, :]
    np.testing.assert_allclose(pkz, g[eng + '_pkz'], rtol=1e-13)


@pytest.mark.parametrize('eng', ENGINES)
def test_sigma8_normalisation(golden, eng):
    """rsigma8 = sigma8 / sigma8_m with sigma8_m from the fftlog sigma_r of the unnormalised P(k) (eisenstein_hu.py:94-103)."""
    g = golden('power')
    c = cosmo_of(g)
    p = ob.derived(h=g['h'], Omega_m=g['Omega_m'], Omega_b=g['Omega_b'], w0_fld=g['w0_fld'], wa_fld=g['wa_fld'])
    D0 = op.growth_factor(np.zeros((8, 1)), p, znorm=0.)[:, 0]
    for i in range(8):
        ci = {n: v[i] for n, v in c.items()}

        def pk(kk):
            return op.pk_z0(kk, eng, sigma8=g['sigma8'][i], n_s=g['n_s'][i], alpha_s=g['alpha_s'][i], **ci) * D0[i]**2

        s8 = np.sqrt(osg.sigma_r2(8., pk))
        np.testing.assert_allclose(g['sigma8'][i] / s8, g[eng + '_rsigma8'][i], rtol=1e-12)
    np.testing.assert_allclose(g[eng + '_sigma8_m'], g['sigma8'], rtol=1e-12)


def eh_default_callable(z):
    p = ob.derived()
    s8 = 0.8

    def pk0(k):
        return op.pk_z0(k, 'eisenstein_hu', sigma8=s8)

    D0 = op.growth_factor(0., p, znorm=0.)
    rs = s8 / np.sqrt(osg.sigma_r2(8., lambda k: pk0(k) * D0**2))
    D = op.growth_factor(np.asarray(z, dtype='f8'), p, znorm=0.)
    return lambda k: (pk0(k) * rs**2)[:, None] * D**2 if np.ndim(z) else pk0(k) * rs**2 * D**2


def test_sigma_eh_callable(golden):
    g = golden('sigma')
    r, z = g['r'], g['z']
    pk = eh_default_callable(z)
    np.testing.assert_allclose(np.sqrt(osg.sigma_r2(r, pk)), g['eh_sigma_rz'], rtol=1e-11)
    np.testing.assert_allclose(np.sqrt(osg.sigma_r2(r[::16], eh_default_callable(z[::8]), method='simpson')), g['eh_sigma_rz_simpson'], rtol=1e-11)
    np.testing.assert_allclose(np.sqrt(osg.sigma_d2(pk)), g['eh_sigma_dz'], rtol=1e-11)
    np.testing.assert_allclose(np.sqrt(osg.sigma_r2(8., pk)), g['eh_sigma8_z'], rtol=1e-11)
    pk1 = eh_default_callable(0.)
    np.testing.assert_allclose(np.sqrt(osg.sigma_r2(r, pk1)), g['eh_sigma_r_1d'], rtol=1e-11)
    np.testing.assert_allclose(np.sqrt(osg.sigma_d2(pk1)), g['eh_sigma_d_1d'], rtol=1e-11)
