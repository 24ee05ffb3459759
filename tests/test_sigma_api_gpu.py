"""The module-level sigma integrals called as the reference's callers call them (interpolator.py:123, 200): ``integrate_sigma_r2(r, pk, kmin, kmax, method,
epsabs, epsrel, nk, kernel)`` / ``integrate_sigma_d2(pk, ...)`` with ``pk`` a callable k (nk,) -> (nk,) or (nk, ncol) -- by position and by keyword --
against the reference's own outputs (tests/golden/sigma_api.npz, `python -m oracle.gen_golden sigma_api`): shapes ``r.shape + pk(kmin).shape``, dtypes,
every method, a custom ``kernel=``.  And the other keyword names VERDICT r5 listed: ``PowerToCorrelation(k=...)``, ``CorrelationToPower(s=...)``,
section getters ``cosmology=``, the sections no engine of this package has."""
import numpy as np
import pytest

from oracle.gen_golden import sigma_api_spectrum, sigma_api_gaussian2

pytestmark = pytest.mark.gpu
# 'quad': the reference's adaptive QUADPACK against this package's refined Simpson rule, each inside max(epsabs, epsrel |I|) of the integral I
# (defaults 1e-5; the results carry I / (2 pi^2), resp. I / (6 pi^2))
RTOL = {'fftlog': 1e-10, 'simpson': 1e-10, 'leggauss': 1e-10, 'quad': 2e-5}
ATOL_QUAD = {'r2': 2e-5 / (2. * np.pi**2), 'd2': 2e-5 / (6. * np.pi**2)}


@pytest.mark.parametrize('ncol', [0, 3])
def test_reference_calling_convention(golden, ncol):
    from cosmoprimo_amd.interpolator import integrate_sigma_r2, integrate_sigma_d2
    g = golden('sigma_api')
    r = g['r']
    pk = lambda k: sigma_api_spectrum(k, ncol)      # noqa: E731

    def check(name, got, rtol):
        ref = g['%s_%d' % (name, ncol)]
        assert isinstance(got, np.ndarray) and got.shape == ref.shape and got.dtype == ref.dtype, (name, got.shape, ref.shape, got.dtype, ref.dtype)
        atol = ATOL_QUAD[name[:2]] if name.endswith('quad') else 0.
        np.testing.assert_allclose(got, ref, rtol=rtol, atol=atol, err_msg=name)

    for method in ('fftlog', 'simpson', 'leggauss', 'quad')[:4 if ncol == 0 else 3]:
        check('r2_' + method, integrate_sigma_r2(r, pk, method=method), RTOL[method])
        check('r2s_' + method, integrate_sigma_r2(8., pk, 1e-7, 1e2, method), RTOL[method])                  # by position, as the reference's signature
    for method in ('simpson', 'leggauss', 'quad')[:3 if ncol == 0 else 2]:
        check('d2_' + method, integrate_sigma_d2(pk, method=method), RTOL[method])
        check('r2g_' + method, integrate_sigma_r2(r, pk, method=method, kernel=sigma_api_gaussian2), RTOL[method])
    check('r2_range', integrate_sigma_r2(r, pk, kmin=1e-5, kmax=10., nk=512, method='simpson'), 1e-10)
    check('r2_range', integrate_sigma_r2(r, pk, 1e-5, 10., 'simpson', 1e-5, 1e-5, 512), 1e-10)                  # epsabs, epsrel, nk in the reference's order
    check('d2_range', integrate_sigma_d2(pk, 1e-5, 10., 'simpson', 1e-5, 1e-5, 512), 1e-10)
    check('r2_f4', integrate_sigma_r2(r.astype('f4'), pk), 1e-6)
    # the kernel is not used by 'fftlog' (interpolator.py:285-288): the top-hat variance whatever it is
    np.testing.assert_array_equal(integrate_sigma_r2(r, pk, kernel=sigma_api_gaussian2), integrate_sigma_r2(r, pk))
    # empty spectra: zeros of shape r.shape + pshape
    empty = integrate_sigma_r2(r, lambda k: np.zeros(np.shape(k) + (0,)))
    assert empty.shape == r.shape + (0,)


def test_keyword_names_of_the_reference():
    import cosmoprimo_amd as cp
    k = np.logspace(-4, 1, 256)
    pk = sigma_api_spectrum(k)
    s1, xi1 = cp.PowerToCorrelation(k=k, ell=0)(pk)
    s2, xi2 = cp.PowerToCorrelation(k)(pk)
    assert np.array_equal(s1, s2) and np.array_equal(xi1, xi2)
    k1, p1 = cp.CorrelationToPower(s=s1, ell=0)(xi1)
    k2, p2 = cp.CorrelationToPower(s1)(xi1)
    assert np.array_equal(k1, k2) and np.array_equal(p1, p2)
    assert cp.FFTlog(k, cp.fftlog.BesselJKernel(0))._engine.name == 'numpy'      # the reference's default engine by name
    cosmo = cp.Cosmology()
    ba = cp.Background(cosmology=cosmo, engine='eisenstein_hu')
    assert ba is cosmo.get_background() and cp.Fourier(cosmology=cosmo) is cosmo.get_fourier()
    assert ba.Omega0_m == cosmo['Omega_m'] and isinstance(type(ba).Omega0_m, property)
    th = cosmo.get_thermodynamics()
    assert th.rs_drag > 0 and isinstance(type(th).rs_drag, property)
    with pytest.raises(AttributeError):
        th.rs_drag = 1.      # read-only, as utils.addproperty makes them
    # sections no analytic engine has: the getters exist and fail as the reference's do for such an engine (cosmology.py:557-571: KeyError)
    for getter in (lambda: cp.Harmonic(cosmo), lambda: cosmo.get_harmonic(), lambda: cosmo.engine.get_harmonic(), lambda: cp.cosmology.Perturbations(cosmo)):
        with pytest.raises(KeyError):
            getter()
    # tabulated interpolators refuse unknown keywords of __call__, callable-built ones hand them to the callable
    interp = cp.PowerSpectrumInterpolator1D(k, pk)
    assert np.isnan(interp(1e5)) and np.isfinite(interp(k[3], bounds_error=True))
    with pytest.raises(ValueError):
        interp(1e5, bounds_error=True)
    with pytest.raises(TypeError):
        interp(k, islogk=True)
    seen = {}

    def callable_pk(kk, **kwargs):
        seen.update(kwargs)
        return sigma_api_spectrum(kk)

    cp.PowerSpectrumInterpolator1D.from_callable(pk_callable=callable_pk)(k, tag=3)
    assert seen == {'tag': 3}


def test_default_order_along_z_does_not_construct():
    """interpolator.py:1262: ``int(interp_order_z)`` of the default None raises; ``to_xi()`` always passes an order."""
    import cosmoprimo_amd as cp
    s = np.geomspace(1e-2, 1e3, 64)
    table = (s[:, None] / 5.)**-1.8 * np.ones(4)
    with pytest.raises(TypeError):
        cp.CorrelationFunctionInterpolator2D(s, np.linspace(0., 1., 4), table)
    cp.CorrelationFunctionInterpolator2D(s, np.linspace(0., 1., 4), table, interp_order_z=3)
