"""The reference's own interpolator test flows (tests/test_interpolator.py: test_power_spectrum :35-121, test_correlation_function :123-165,
test_extrap_1d :168-229, test_extrap_2d :232-300, test_nan :328-337) exercised through this package's API with the same assertions:
shapes, dtypes, ordering, clone / from_callable / to_1d / to_xi / to_pk round trips, bounds errors and NaN rules."""
import warnings

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def cp():
    import torch
    assert torch.cuda.is_available(), 'GPU tests need a ROCm device'
    import cosmoprimo_amd
    return cosmoprimo_amd


def check_shape_1d(interp, shape=()):
    assert interp(0.1).shape == shape
    assert interp([]).shape == (0,) + shape
    assert interp([[0.1, 0.2]] * 3).shape == (3, 2) + shape
    assert interp(np.array([[0.1, 0.2]] * 3, dtype='f4')).dtype.itemsize == 4
    assert np.allclose(interp([0.2, 0.1]), interp([0.1, 0.2])[::-1], atol=0)


def check_shape_2d(interp, grid=True):
    assert interp(0.1, 0.1).shape == ()
    if grid:
        assert interp(np.array([]), np.array(0.1)).shape == (0, )
        assert interp([], []).shape == (0, 0)
        assert interp(0.1, [0.1, 0.1]).shape == (2, )
        assert interp([[0.1, 0.2]] * 3, 0.1).shape == (3, 2)
        assert interp([[0.1, 0.2]] * 3, [0.1]).shape == (3, 2, 1)
        assert interp([[0.1, 0.2]] * 3, [[0.1, 0.1, 0.2]] * 3).shape == (3, 2, 3, 3)
        assert interp(np.array([[0.1, 0.2]] * 3, dtype='f4'), np.array(0.1, dtype='f4')).dtype.itemsize == 4
        assert np.allclose(interp([0.2, 0.1], [0.1, 0.]), interp([0.1, 0.2], [0., 0.1])[::-1, ::-1], atol=0)
    else:
        assert interp([], [], grid=False).shape == (0, )
        assert interp([0.1, 0.2], [0.1, 0.2], grid=False).shape == (2, )
        assert interp([[0.1, 0.2]] * 3, [[0.1, 0.2]] * 3, grid=False).shape == (3, 2)
        assert np.allclose(interp([0.2, 0.1], [0.1, 0.], grid=False), interp([0.1, 0.2], [0., 0.1], grid=False)[::-1], atol=0)


def test_power_spectrum(cp):
    warnings.simplefilter('ignore')
    cosmo = cp.Cosmology()
    tr = cp.Transfer(cosmo, engine='eisenstein_hu')
    k = np.logspace(-3, 1.5, 100)
    pk = tr.transfer_k(k)**2 * k ** cosmo['n_s']
    interp = cp.PowerSpectrumInterpolator1D(k, pk)
    check_shape_1d(interp)
    interp2d = cp.PowerSpectrumInterpolator2D(k, z=[0., 0.5, 1., 1.5], pk=np.repeat(pk[:, None], 4, axis=-1))
    interp2d(k, z=0.)
    interp2 = interp.clone()
    assert np.all(interp2(np.ones((4, 2))) == interp(np.ones((4, 2))))
    check_shape_1d(interp.sigma_r)

    interp = cp.PowerSpectrumInterpolator2D(k, z=0, pk=pk, growth_factor_sq=lambda z: np.ones_like(z))
    assert np.allclose(interp(k, z=np.random.uniform(0., 1., 10)), pk[:, None], atol=0, rtol=1e-5)
    check_shape_2d(interp)
    check_shape_2d(interp, grid=False)
    interp2 = interp.clone()
    assert np.all(interp2._pk == interp._pk)
    assert np.allclose(interp2(k, z=[0] * 2), interp(k, z=[0] * 2), atol=1e-18, rtol=1e-18)

    rng = np.random.RandomState(seed=42)
    z = np.linspace(1., 0., 10)
    interp = cp.PowerSpectrumInterpolator2D(k, z=z, pk=np.array([pk] * len(z)).T)
    check_shape_2d(interp)
    assert np.allclose(interp(k, z=rng.uniform(0., 1., 10)), pk[:, None], atol=0, rtol=1e-5)
    check_shape_1d(interp.sigma8_z)
    check_shape_1d(interp.sigma_dz)
    check_shape_2d(interp.sigma_rz)
    interp = cp.PowerSpectrumInterpolator2D(k, z=z, pk=np.array([pk * (iz + 1) / len(z) for iz in range(len(z))]).T)
    check_shape_2d(interp.growth_rate_rz)
    dz = 1e-3
    assert np.allclose(interp.growth_rate_rz(8., dz * 2., dz=dz), interp.growth_rate_rz(8., 0., dz=dz), rtol=1e-2)
    interp = cp.PowerSpectrumInterpolator2D(k, z=z, pk=np.array([pk] * len(z)).T, extrap_kmin=1e-6, extrap_kmax=1e2)
    check_shape_2d(interp)

    for engine in ['eisenstein_hu', 'eisenstein_hu_nowiggle_variants']:
        fo = cp.Fourier(cosmo, engine=engine)
        interp = fo.pk_interpolator()
        k = np.logspace(-4, 2, 100)
        z = np.linspace(0, 4, 10)
        check_shape_2d(interp)
        pk = interp(k, z)
        interp2 = interp.clone()
        assert np.allclose(interp2(k, z), pk, rtol=1e-4)
        interp2 = interp.clone(pk=2 * interp.pk)
        assert np.allclose(interp2(k, z), 2 * pk, rtol=1e-4)
        for iz, zz in enumerate(z):
            interp1d = interp.to_1d(z=zz)
            check_shape_1d(interp1d)
            assert np.allclose(interp1d.extrap_kmin, interp.extrap_kmin)
            assert np.allclose(interp1d.extrap_kmax, interp.extrap_kmax)
            assert np.allclose(interp1d(k), pk[:, iz], rtol=2e-5)
            assert np.allclose(interp.sigma8_z(zz), interp.to_1d(zz).sigma8(), rtol=1e-4)
            assert np.allclose(interp.sigma_dz(zz), interp.to_1d(zz).sigma_d(), rtol=1e-4)
            assert np.allclose(interp.sigma_dz(zz, nk=None), interp.to_1d(zz).sigma_d(), rtol=1e-4)
        interp2 = cp.PowerSpectrumInterpolator2D.from_callable(interp.k, interp.z, interp)
        check_shape_2d(interp2)
        check_shape_2d(interp2, grid=False)
        assert np.allclose(interp2(k, z), interp(k, z), rtol=1e-4)
        interp_1d = interp.to_1d(z=0.)
        check_shape_1d(interp_1d)
        interp_1d2 = interp_1d.from_callable(interp_1d.k, interp_1d)
        check_shape_1d(interp_1d2)
        assert np.allclose(interp_1d2(k), interp_1d(k), rtol=1e-4)

        k = np.logspace(-4, 2, 1000)
        z = np.linspace(0, 4, 10)
        pk_interp = fo.pk_interpolator()
        pk_interp = cp.PowerSpectrumInterpolator2D(k, z, pk_interp(k, z))
        pk_interp_1d = pk_interp.to_1d(z=z)
        assert np.allclose(pk_interp_1d(k), pk_interp(k, z))
        check_shape_1d(pk_interp_1d, shape=(len(z),))
        xi_interp_1d = pk_interp_1d.to_xi()
        s = xi_interp_1d.s
        assert np.allclose(xi_interp_1d(s), pk_interp.to_xi()(s, z))


def test_correlation_function(cp):
    warnings.simplefilter('ignore')
    cosmo = cp.Cosmology()
    for engine in ['eisenstein_hu', 'eisenstein_hu_nowiggle_variants']:
        fo = cp.Fourier(cosmo, engine=engine)
        pk_interp = fo.pk_interpolator()
        xi_interp = pk_interp.clone(extrap_kmin=1e-5, extrap_kmax=1e2).to_xi()
        pk_interp2 = xi_interp.to_pk()
        s = np.logspace(-2, 2, 100)
        z = np.linspace(0, 4, 10)
        assert np.allclose(xi_interp.clone()(s, z), xi_interp(s, z), rtol=1e-4)
        check_shape_2d(xi_interp)
        xi_interp2 = cp.CorrelationFunctionInterpolator2D.from_callable(xi_interp.s, xi_interp.z, xi_interp)
        check_shape_2d(xi_interp2)
        assert np.allclose(xi_interp2(s, z), xi_interp(s, z), rtol=1e-4)
        xi_interp_1d = xi_interp.to_1d(z=0.)
        check_shape_1d(xi_interp_1d)
        xi_interp_1d2 = xi_interp_1d.from_callable(xi_interp_1d.s, xi_interp_1d)
        check_shape_1d(xi_interp_1d2)
        assert np.allclose(xi_interp_1d2(s), xi_interp_1d(s), rtol=1e-4)
        k = np.logspace(-4, 1, 100)
        assert np.allclose(pk_interp(k, z), pk_interp2(k, z), rtol=1e-2)
        for zz in z[1:]:
            pk1 = fo.pk_interpolator().to_1d(z=zz)
            pk2 = pk1.clone(extrap_kmin=1e-5, extrap_kmax=1e2).to_xi().to_pk()
            assert np.allclose(pk1(k), pk2(k), rtol=1e-2)
            pk2 = pk1.clone(extrap_kmin=1e-5, extrap_kmax=1e2).to_xi().clone().to_pk()
            assert np.allclose(pk1(k), pk2(k), rtol=1e-2)
            assert np.allclose(xi_interp.sigma_dz(zz), pk1.sigma_d(), rtol=1e-4)
            assert np.allclose(xi_interp.sigma8_z(zz), pk1.sigma8(), rtol=1e-4)
            assert np.allclose(xi_interp.sigma8_z(zz), xi_interp.to_1d(zz).sigma8(), rtol=1e-4)


def test_extrap(cp):
    warnings.simplefilter('ignore')
    cosmo = cp.Cosmology()
    fo = cp.Fourier(cosmo, engine='eisenstein_hu')
    k = np.logspace(-4, 2, 1000)
    k_extrap = np.logspace(-6, 3, 1000)
    k_eval = k_extrap[1:-1]
    # 1D (reference test_extrap_1d)
    pk_c = fo.pk_interpolator(k=k_extrap, extrap_kmin=k_extrap[0], extrap_kmax=k_extrap[-1]).to_1d(z=0.)
    pk_t = cp.PowerSpectrumInterpolator1D(k, pk_c(k), extrap_kmin=k_extrap[0], extrap_kmax=k_extrap[-1])
    assert np.allclose(pk_t(k), pk_c(k), atol=0, rtol=0.1) and np.allclose(pk_t(k_eval), pk_c(k_eval), atol=0, rtol=0.1)
    assert np.allclose(pk_t.extrap_kmin, pk_c.k[0]) and np.allclose(pk_t.extrap_kmax, pk_c.k[-1])
    pk_c(k_eval / 2., bounds_error=False)
    pk_t(k_eval, bounds_error=True)
    assert np.isnan(pk_t(k_eval[0] / 2.)) and np.isnan(pk_t(k_eval[-1] * 2))
    for bad in (k_eval / 2., k_eval * 2.):
        with pytest.raises(ValueError):
            pk_t(bad, bounds_error=True)
    xi_c, xi_t = pk_c.to_xi(), pk_t.to_xi()
    s_eval = xi_t.s
    xi_t(s_eval, bounds_error=True)
    assert np.isnan(xi_t(s_eval[0] / 2., bounds_error=False)) and np.isnan(xi_t(s_eval[-1] * 2., bounds_error=False))
    for bad in (s_eval / 2., s_eval * 2.):
        with pytest.raises(ValueError):
            xi_t(bad, bounds_error=True)
    assert np.allclose(xi_t(s_eval), xi_c(s_eval), rtol=0.1)
    assert np.allclose(xi_t.to_pk()(k), pk_c(k), atol=0, rtol=1e-2)
    # 2D (reference test_extrap_2d)
    z = np.linspace(0, 4, 10)
    pk_c = fo.pk_interpolator(k=k_extrap, z=z, extrap_kmin=k_extrap[0], extrap_kmax=k_extrap[-1])
    pk_t = cp.PowerSpectrumInterpolator2D(k, z, pk_c(k, z), extrap_kmin=k_extrap[0], extrap_kmax=k_extrap[-1])
    assert np.allclose(pk_t(k, z), pk_c(k, z), atol=0, rtol=0.1) and np.allclose(pk_t(k_eval, z), pk_c(k_eval, z), atol=0, rtol=0.1)
    assert np.isnan(pk_t(k_eval[0] / 2., z, bounds_error=False)).all() and np.isnan(pk_t(k_eval[-1] * 2., z, bounds_error=False)).all()
    for args in ((k_eval / 2., z), (k_eval * 2., z), (k_eval, z * 2.)):
        with pytest.raises(ValueError):
            pk_t(*args, bounds_error=True)
    assert np.isnan(pk_t(k_eval, z[-1] * 2., bounds_error=False)).all()
    xi_c, xi_t = pk_c.to_xi(), pk_t.to_xi()
    s_eval = xi_t.s
    xi_t(s_eval, z)
    assert np.isnan(xi_t(s_eval[0] / 2., z, bounds_error=False)).all() and np.isnan(xi_t(s_eval[-1] * 2., z, bounds_error=False)).all()
    for args in ((s_eval / 2., z), (s_eval * 2., z), (s_eval, z * 2.)):
        with pytest.raises(ValueError):
            xi_t(*args, bounds_error=True)
    assert np.isnan(xi_t(s_eval, z[-1] * 2., bounds_error=False)).all()
    assert np.allclose(xi_t(s_eval, z), xi_c(s_eval, z), rtol=0.1)
    assert np.allclose(xi_t.to_pk()(k, z=0.), pk_c(k, z=0.), atol=0, rtol=1e-2)


def test_nan(cp):
    k = np.logspace(-4, 2, 1000)
    pk = k**2
    pk[:2] *= -1
    assert np.isnan(cp.PowerSpectrumInterpolator1D(k, pk)(k)).all()
    z = np.linspace(0., 2., 4)
    assert np.isnan(cp.PowerSpectrumInterpolator2D(k, z, pk[..., None][..., [0] * len(z)])(k, z=1.)).all()
