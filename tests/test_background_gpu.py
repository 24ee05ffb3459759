"""GPU parity of the background distance kernel against the oracle and the reference's golden vectors (G8).
Tolerance: pointwise relative <= 1e-10; measured ~1e-15."""
import numpy as np
import pytest

from oracle import background as ob

pytestmark = pytest.mark.gpu
RTOL = 1e-10
NAMES = ['comoving_radial_distance', 'comoving_transverse_distance', 'angular_diameter_distance', 'luminosity_distance']


@pytest.fixture(scope='module')
def bg():
    import torch
    assert torch.cuda.is_available()
    from cosmoprimo_amd import background
    return background


def gparams(g):
    return {k: g[k] for k in ['h', 'Omega_b', 'Omega_k', 'w0_fld', 'wa_fld']}


def test_golden_batch(bg, golden):
    g = golden('background')
    z = g['z']
    for name in NAMES:
        out = bg.distance(name, z, gparams(g), Omega_m=g['Omega_m'])
        assert out.shape == (32, z.size)
        ref = g[name]
        m = ref != 0
        assert np.abs(out[m] / ref[m] - 1).max() < RTOL, name
        assert np.all(out[~m] == 0)
    e = bg.distance('efunc', z, gparams(g), Omega_m=g['Omega_m'])
    assert np.abs(e / g['efunc'] - 1).max() < 1e-13
    hub = bg.distance('hubble_function', z, gparams(g), Omega_m=g['Omega_m'])
    assert np.abs(hub / (g['efunc'] * 100 * g['h'][:, None]) - 1).max() < 1e-13


def test_scalar_cosmology_contracts(bg, golden):
    """shape / dtype / NaN rules of utils.flatarray + Interpolator1D (reference utils.py:98-138, jax.py:187-196)."""
    g = golden('background')
    p = {k: float(g[k][0]) for k in ['h', 'Omega_b', 'Omega_k', 'w0_fld', 'wa_fld']}
    om = float(g['Omega_m'][0])
    out = bg.distance('comoving_radial_distance', np.array([-0.1, 1e4, np.nan]), p, Omega_m=om)
    assert np.isnan(out).all()
    out = bg.distance('comoving_radial_distance', np.linspace(0., 2., 5).astype('f4'), p, Omega_m=om)
    assert out.dtype == np.float32 and np.allclose(out, g['f4'], rtol=1e-6)
    assert bg.distance('comoving_radial_distance', 0.5, p, Omega_m=om).shape == ()
    assert bg.distance('comoving_radial_distance', np.zeros((0,)), p, Omega_m=om).shape == (0,)
    z = np.linspace(0., 3., 24).reshape(2, 3, 4)
    out = bg.distance('luminosity_distance', z, p, Omega_m=om)
    assert out.shape == z.shape
    ref = ob.distances(z, ob.derived(Omega_m=om, **p))['luminosity_distance']
    assert np.abs(out[ref > 0] / ref[ref > 0] - 1).max() < RTOL
    with pytest.raises(ValueError):
        bg.distance('nope', z, p)


def test_config5_samples_vs_oracle(bg):
    """BASELINE config 5 generator (Omega_m, w0, wa, z per sample), per-sample z, torch in/out."""
    import torch
    rng = np.random.default_rng(3)
    n = 4096
    om, w0, wa, z = rng.uniform(0.1, 0.5, n), rng.uniform(-1.5, -0.5, n), rng.uniform(-1., 0.5, n), rng.uniform(0., 3., n)
    out = bg.distance('comoving_radial_distance', torch.as_tensor(z, device='cuda')[:, None],
                      dict(w0_fld=torch.as_tensor(w0, device='cuda'), wa_fld=torch.as_tensor(wa, device='cuda')),
                      Omega_m=torch.as_tensor(om, device='cuda'), per_cosmology_z=True)
    assert isinstance(out, torch.Tensor) and out.shape == (n, 1)
    ref = ob.comoving_radial_distance(z[:, None], ob.derived(Omega_m=om, w0_fld=w0, wa_fld=wa))
    assert np.abs(out.cpu().numpy() / ref - 1).max() < RTOL


def test_every_interval_and_knot(bg):
    """z on every knot, at every interval midpoint and just inside both ends (all 118 elimination split points)."""
    zc = ob.z_knots()
    z = np.concatenate([zc, 0.5 * (zc[1:] + zc[:-1]), [1e-12, zc[-1] * (1 - 1e-12)]])
    for pars in [dict(), dict(Omega_m=0.2, w0_fld=-0.7, wa_fld=-0.8, Omega_k=0.05), dict(Omega_m=0.45, w0_fld=-1.3, wa_fld=0.4, Omega_k=-0.08, h=0.6)]:
        om = pars.pop('Omega_m', None)
        ref = ob.distances(z, ob.derived(Omega_m=om, **pars))
        for name in NAMES:
            out = bg.distance(name, z, pars, Omega_m=om)
            m = ref[name] != 0
            assert np.abs(out[m] / ref[name][m] - 1).max() < RTOL, (name, pars)


def test_densities(golden):
    """a19: every BaseBackground density, density parameter and T_cmb(z) through the device kernel (new kinds of cp_background_distance)."""
    import warnings
    import cosmoprimo_amd as cp
    from oracle.gen_golden import DENSITY_NAMES, DENSITY_PARAMS
    g = golden('densities')
    z = g['z']
    for i, par in enumerate(DENSITY_PARAMS):
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            ba = cp.Cosmology(engine='eisenstein_hu', **par).get_background()
        for name in DENSITY_NAMES:
            out = getattr(ba, name)(z)
            assert out.shape == z.shape and out.dtype == np.float64
            np.testing.assert_allclose(out, g['c%d_%s' % (i, name)], rtol=1e-13, atol=1e-300, err_msg='%d %s' % (i, name))
        assert ba.rho_g(z.astype('f4')).dtype == np.float32 and ba.Omega_k(0.).shape == ()
        # time / age (cosmology.py:2000-2025) on the 400-knot grid: cancellation (T_last - T(z)) costs digits at high z
        np.testing.assert_allclose(ba.time(z), g['c%d_time' % i], rtol=1e-9)
        np.testing.assert_allclose(ba.time(z[:8]), g['c%d_time' % i][:8], rtol=1e-12)
        np.testing.assert_allclose(ba.age, g['c%d_age' % i], rtol=1e-13)
        assert np.ndim(ba.age) == 0 and np.isnan(ba.time(np.array([-0.1, 1e8]))).all()
        # sound horizon: fixed-depth Romberg, one wave per sample (cosmology.py:1914-1933); theta_cosmomc (:202-228, 404-408)
        from oracle.gen_golden import RS_Z
        np.testing.assert_allclose(ba.rs(np.array(RS_Z)), g['c%d_rs' % i], rtol=1e-10)
        cosmo_i = cp.Cosmology(engine='eisenstein_hu', **par)
        np.testing.assert_allclose(cosmo_i['theta_cosmomc'], g['c%d_theta_cosmomc' % i], rtol=1e-10)
        np.testing.assert_allclose(cosmo_i['theta_MC_100'], 100. * g['c%d_theta_cosmomc' % i], rtol=1e-10)
        with pytest.raises(cp.CosmologyComputationError):
            ba.rs(0.)        # the reference fails there as well: its rule misses the 1e-7 tolerance
        # linear growth from its ODE (DefaultBackground, cosmology.py:2044-2093); the analytic engines' own Background keeps the CPT92 form
        from cosmoprimo_amd.cosmology import DefaultBackground
        bd = DefaultBackground(cosmo_i.engine)
        zg = g['zg']
        np.testing.assert_allclose(bd.growth_factor(zg), g['c%d_growth_factor_ode' % i], rtol=1e-10)
        np.testing.assert_allclose(bd.growth_factor(zg, znorm=10.), g['c%d_growth_factor_ode_znorm' % i], rtol=1e-10)
        np.testing.assert_allclose(bd.growth_factor(zg, mass='cb'), g['c%d_growth_factor_ode_cb' % i], rtol=1e-10)
        np.testing.assert_allclose(bd.growth_rate(zg), g['c%d_growth_rate_ode' % i], rtol=1e-10)
        np.testing.assert_allclose(bd.growth_factor(zg), ob.growth_factor_ode(zg, ob.derived(**{k: v for k, v in par.items() if k != 'cs2_fld'},
                                                                                             **({'Omega_cdm': 0.25} if 'Omega_m' not in par else {}))), rtol=1e-10)
        assert bd.growth_factor(0.).shape == () and np.isnan(bd.growth_factor(np.array([500.]))).all() and abs(bd.growth_factor(0.) - 1.) < 1e-15
        assert abs(ba.growth_factor(1.) / bd.growth_factor(1.) - 1.) < 0.1       # CPT92 closed form vs ODE: a few % with curvature or w0-wa
    # a batch of cosmologies: leading axis
    ba = cp.Cosmology(engine='eisenstein_hu', Omega_m=np.array([0.3, 0.36]), h=np.array([0.7, 0.64])).get_background()
    out = ba.Omega_cdm(z)
    assert out.shape == (2, z.size)
    np.testing.assert_allclose(out[1], cp.Cosmology(engine='eisenstein_hu', Omega_m=0.36, h=0.64).get_background().Omega_cdm(z), rtol=1e-14)
    np.testing.assert_allclose(ba.age, [cp.Cosmology(engine='eisenstein_hu', Omega_m=om, h=h).get_background().age for om, h in [(0.3, 0.7), (0.36, 0.64)]],
                               rtol=1e-14)
    kn = np.empty(400)
    from cosmoprimo_amd import _lib
    _lib.check(_lib.load().cp_background_knots(_lib.as_double_p(kn), 400))
    np.testing.assert_allclose(kn, g['time_knots'], rtol=1e-15)
    both = cp.Cosmology(engine='eisenstein_hu', Omega_m=np.array([0.3, 0.36]), h=np.array([0.7, 0.64]))
    np.testing.assert_allclose(both['theta_cosmomc'], [cp.Cosmology(engine='eisenstein_hu', Omega_m=om, h=h)['theta_cosmomc'] for om, h in [(0.3, 0.7), (0.36, 0.64)]],
                               rtol=1e-13)
    from cosmoprimo_amd.cosmology import DefaultBackground
    gb = DefaultBackground(both.engine).growth_factor(g['zg'])
    assert gb.shape == (2, g['zg'].size)
    np.testing.assert_allclose(gb[1], DefaultBackground(cp.Cosmology(engine='eisenstein_hu', Omega_m=0.36, h=0.64).engine).growth_factor(g['zg']), rtol=1e-13)


def test_catalogue_path():
    """One cosmology (a few redshifts or a catalogue): the distance table is built once and splined per point (cp_spline_points) -- the same
    natural spline the kernel evaluates per sample, so the same numbers; NaN outside the knots, dtype and container of the input kept."""
    import torch
    from cosmoprimo_amd.cosmology import BaseBackground
    from cosmoprimo_amd.fiducial import DESI
    rng = np.random.default_rng(12)
    for cosmo in (cp_mod().Cosmology(engine='eisenstein_hu', Omega_k=0.05, w0_fld=-0.9, wa_fld=0.1), DESI()):
        ba = cosmo.get_background()
        z = np.concatenate([rng.uniform(0., 5., 50000), 10.**rng.uniform(-6, 3.9, 5000), [0., -0.1, 2e4]])
        fast = ba.comoving_radial_distance(z)
        zt = torch.as_tensor(z[:40000], device='cuda:0', dtype=torch.float32)
        BaseBackground._use_table_spline = False
        try:
            slow = ba.comoving_radial_distance(z)
        finally:
            BaseBackground._use_table_spline = True
        for name in ('angular_diameter_distance', 'comoving_transverse_distance', 'luminosity_distance'):      # through the same table
            quick = getattr(ba, name)(z)
            BaseBackground._use_table_spline = False
            try:
                np.testing.assert_allclose(quick, getattr(ba, name)(z), rtol=1e-11, atol=1e-12 * np.nanmax(slow), equal_nan=True, err_msg=name)
            finally:
                BaseBackground._use_table_spline = True
            assert getattr(ba, name)(np.float32(0.5)).dtype == np.float32 and getattr(ba, name)(zt).dtype == torch.float32
        few = ba.comoving_radial_distance(z[:7])
        assert few.shape == (7,) and np.allclose(few, slow[:7], rtol=1e-11) and ba.comoving_radial_distance(0.5).shape == ()
        assert np.array_equal(np.isnan(fast), np.isnan(slow)) and np.isnan(fast[-2:]).all() and fast[-3] == 0.
        np.testing.assert_allclose(fast, slow, rtol=1e-11, atol=1e-12 * np.nanmax(slow), equal_nan=True)
        out = ba.comoving_radial_distance(zt)
        assert out.is_cuda and out.dtype == torch.float32 and out.shape == zt.shape
        out = ba.comoving_radial_distance(z[:55000].reshape(5, 11000).astype('f4'))
        assert out.dtype == np.float32 and out.shape == (5, 11000)


def cp_mod():
    import cosmoprimo_amd
    return cosmoprimo_amd


def test_distances_extreme_parameters():
    """E(z) and D_C far from the fiducial region against the oracle (the quadrature takes 1 / E from rsqrt on tabulated log(1 + z)): curvature,
    phantom / thawing dark energy, low and high matter density."""
    import torch
    from cosmoprimo_amd import background
    from oracle import background as ob
    rng = np.random.default_rng(22)
    nb = 400
    om, ok = 10.**rng.uniform(-1.5, 0., nb), rng.uniform(-0.3, 0.3, nb)
    w0, wa, h = rng.uniform(-2.5, -0.34, nb), rng.uniform(-2., 0.6, nb), rng.uniform(0.3, 1.5, nb)
    wa = np.where(w0 + wa >= 0.3, 0.25 - w0, wa)      # w(a -> 0) < 1/3
    z = 10.**rng.uniform(-3., 3.5, (nb, 6))
    dev = torch.device('cuda', 0)
    t = lambda v: torch.as_tensor(v, device=dev)      # noqa: E731
    params = dict(w0_fld=t(w0), wa_fld=t(wa), Omega_k=t(ok), h=t(h))
    for kind in ('efunc', 'comoving_radial_distance'):
        out = background.distance(kind, t(z), params, Omega_m=t(om), per_cosmology_z=True).cpu().numpy()
        assert np.isfinite(out).mean() > 0.9
        for i in range(0, nb, 3):      # (closed models with little matter have E^2 < 0 somewhere: NaN there in the reference as well)
            p = ob.derived(Omega_m=om[i], Omega_k=ok[i], w0_fld=w0[i], wa_fld=wa[i], h=h[i])
            with np.errstate(all='ignore'):
                try:
                    ref = ob.efunc(z[i], p) if kind == 'efunc' else ob.comoving_radial_distance(z[i], p)
                except ValueError:      # scipy refuses the NaN table of such a model (so does the reference): every distance is NaN here
                    assert np.isnan(out[i]).all()
                    continue
            np.testing.assert_allclose(out[i], ref, rtol=1e-10, equal_nan=True, err_msg='%s %d' % (kind, i))
