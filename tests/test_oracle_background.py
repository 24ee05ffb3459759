"""Pin the numpy background oracle against golden vectors from the reference (G8)."""
import numpy as np

from oracle import background as ob


def params_of(g):
    return ob.derived(h=g['h'], Omega_m=g['Omega_m'], Omega_b=g['Omega_b'], Omega_k=g['Omega_k'], w0_fld=g['w0_fld'], wa_fld=g['wa_fld'])


def test_knots_and_table(golden):
    g = golden('background')
    p = params_of(g)
    zc, tab = ob.distance_table(p)
    assert zc.size == 119 and np.array_equal(zc, g['zc'])
    np.testing.assert_allclose(tab, g['table'], rtol=1e-15, atol=0)


def test_efunc_and_distances(golden):
    g = golden('background')
    p = params_of(g)
    z = g['z']
    np.testing.assert_allclose(ob.efunc(z[None, :], {k: v[:, None] for k, v in p.items()}), g['efunc'], rtol=1e-15)
    d = ob.distances(np.broadcast_to(z, (32, z.size)), p)
    for name in ['comoving_radial_distance', 'comoving_transverse_distance', 'angular_diameter_distance', 'luminosity_distance']:
        np.testing.assert_allclose(d[name], g[name], rtol=1e-14, atol=1e-12)
    out = ob.comoving_radial_distance(np.array([-0.1, 1e4]), {k: v[0] for k, v in p.items()})
    assert np.isnan(out).all() and np.isnan(g['nan_outside']).all()


def test_closed_form_lcdm_flat_matter_only():
    # Einstein-de Sitter-like check of the quadrature + spline: E = (1+z)^1.5 when Omega_m = 1 -> D_C = 2 c/H0 (1 - 1/sqrt(1+z))
    p = ob.derived(h=0.7, Omega_cdm=0.95, Omega_b=0.05, T_cmb=1e-8, N_ur=0.)
    z = np.linspace(0.2, 5., 20)
    dc = ob.comoving_radial_distance(z, p)
    np.testing.assert_allclose(dc, 2 * ob.C_KMS / 100. * (1 - 1 / np.sqrt(1 + z)), rtol=1e-5)


def test_densities(golden):
    """a19: BaseBackground rho_x / Omega_x / T_cmb(z) (oracle/background.py: densities) against the reference, 5 cosmologies."""
    from oracle.gen_golden import DENSITY_NAMES, DENSITY_PARAMS
    g = golden('densities')
    z = g['z']
    for i, par in enumerate(DENSITY_PARAMS):
        par = dict(par)
        cs2 = par.pop('cs2_fld', 1.)
        T_cmb = par.get('T_cmb', ob.TCMB)
        p = ob.derived(**({'Omega_cdm': 0.25} if 'Omega_m' not in par else {}), **par)
        has_fld = bool(g['c%d_has_fld' % i])
        assert has_fld == ((par.get('w0_fld', -1.) != -1.) or (par.get('wa_fld', 0.) != 0.) or (cs2 != 1.))
        d = ob.densities(z, p, T_cmb=T_cmb, has_fld=has_fld)
        for name in DENSITY_NAMES:
            ref = g['c%d_%s' % (i, name)]
            if name in ('rho_ncdm_tot', 'p_ncdm_tot'):
                assert (ref == 0.).all()
                continue
            np.testing.assert_allclose(d[name], ref, rtol=1e-13, atol=1e-300, err_msg='%d %s' % (i, name))
        # time / age: DefaultBackground.time, .age (cosmology.py:2000-2025)
        np.testing.assert_allclose(ob.time(z, p), g['c%d_time' % i], rtol=1e-11)
        np.testing.assert_allclose(ob.age(p), g['c%d_age' % i], rtol=1e-13)
        # sound horizon (fixed-depth Romberg): rs(z) and theta_cosmomc (cosmology.py:202-228, 404-408, 1914-1933)
        from oracle.gen_golden import RS_Z
        np.testing.assert_allclose([ob.rs(zz, p) for zz in RS_Z], g['c%d_rs' % i], rtol=1e-11)
        h2 = p['h']**2
        zstar = ob.zstar_cosmomc(p['Omega_b'] * h2, (p['Omega_cdm'] + p['Omega_b']) * h2)
        theta = ob.rs(zstar, p, cosmomc=True) * p['h'] / ob.distances(np.array([zstar]), p)['comoving_transverse_distance'][0]
        np.testing.assert_allclose(theta, g['c%d_theta_cosmomc' % i], rtol=1e-10)
        # ODE growth of DefaultBackground (cosmology.py:2044-2093)
        zg = g['zg']
        np.testing.assert_allclose(ob.growth_factor_ode(zg, p), g['c%d_growth_factor_ode' % i], rtol=1e-11)
        np.testing.assert_allclose(ob.growth_factor_ode(zg, p, znorm=10.), g['c%d_growth_factor_ode_znorm' % i], rtol=1e-11)
        np.testing.assert_allclose(ob.growth_factor_ode(zg, p, mass='cb'), g['c%d_growth_factor_ode_cb' % i], rtol=1e-11)
        np.testing.assert_allclose(ob.growth_rate_ode(zg, p), g['c%d_growth_rate_ode' % i], rtol=1e-11)
    np.testing.assert_allclose(ob.time_knots(), g['time_knots'], rtol=1e-15)
    assert np.isnan(g['time_nan_outside']).all()
