"""GPU parity of the xi side: CorrelationFunctionInterpolator1D/2D, to_xi / to_pk round
trips and sigma through to_pk, against golden vectors generated from the reference (tests/golden/xi.npz) and the oracle.
Tolerances: tilted-space 1e-12 for FFTLog outputs; 1e-9 pointwise for quantities that went through two FFTLogs and two
log-space splines (conditioning of 10**spline(log P))."""
import numpy as np
import pytest

from oracle import interp as oi

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def cp():
    import torch
    assert torch.cuda.is_available(), 'GPU tests need a ROCm device'
    import cosmoprimo_amd
    return cosmoprimo_amd


def tilted(a, b, s, q=1.5):
    w = s**q if np.ndim(a) == 1 else s[:, None]**q
    assert (np.isnan(a) == np.isnan(b)).all()
    return np.nanmax(np.abs((a - b) * w)) / np.nanmax(np.abs(b * w))


def test_xi_1d(cp, golden):
    g = golden('xi')
    sq, kq = g['sq'], g['kq']
    interp = cp.Cosmology(engine='eisenstein_hu').get_fourier().pk_interpolator()
    i1 = interp.to_1d(z=0.)
    xi1 = i1.to_xi()
    assert isinstance(xi1, cp.CorrelationFunctionInterpolator1D)
    np.testing.assert_allclose(xi1.s, g['xi1_s'], rtol=1e-13)
    assert tilted(xi1.xi, g['xi1_xi'], xi1.s) < 1e-13
    assert tilted(xi1(sq), g['xi1_eval'], sq) < 1e-12
    oob = xi1(np.array([xi1.smin * 0.5, 1., xi1.smax * 2.]))
    assert np.isnan(oob[[0, 2]]).all() and abs(oob[1] / g['xi1_eval_oob'][1] - 1.) < 1e-10
    with pytest.raises(ValueError):
        xi1(np.array([xi1.smin * 0.5]), bounds_error=True)
    # default k range: P(k) back rings below zero at the edges -> NaN everywhere, as in the reference
    assert np.isnan(xi1.to_pk()(kq)).all() and np.isnan(xi1.sigma8())
    # the reference's round trip (tests/test_interpolator.py:123-165): narrowed k range
    c1 = i1.clone(extrap_kmin=1e-5, extrap_kmax=1e2)
    np.testing.assert_allclose(c1.k, g['c1_k'], rtol=1e-14)
    np.testing.assert_allclose(c1.pk, g['c1_pk'], rtol=1e-10)
    xc1 = c1.to_xi()
    np.testing.assert_allclose(xc1.s, g['xc1_s'], rtol=1e-13)
    assert tilted(xc1.xi, g['xc1_xi'], xc1.s) < 1e-11
    pc1 = xc1.to_pk()
    assert isinstance(pc1, cp.PowerSpectrumInterpolator1D)
    np.testing.assert_allclose(pc1.k, g['xc1_to_pk_k'], rtol=1e-13)
    assert tilted(pc1.pk, g['xc1_to_pk_pk'], pc1.k) < 1e-9
    np.testing.assert_allclose(pc1(kq), g['xc1_to_pk_eval'], rtol=1e-9)
    np.testing.assert_allclose(xc1.sigma8(), g['xc1_sigma8'], rtol=1e-9)
    np.testing.assert_allclose(xc1.sigma_r(np.array([2., 8., 30.])), g['xc1_sigma_r'], rtol=1e-9)
    np.testing.assert_allclose(xc1.sigma_d(), g['xc1_sigma_d'], rtol=1e-9)
    np.testing.assert_allclose(pc1(kq), i1(kq), rtol=1e-2)      # the round trip itself (reference test: rtol 1e-2)
    # contracts (reference tests/test_interpolator.py:8-32): dtype and shape follow the input
    assert xi1(sq[:5].astype('f4')).dtype == np.float32 and xi1(1.).shape == () and xi1(np.zeros((0,))).shape == (0,)
    # rescale_sigma8
    xc1.rescale_sigma8(0.5)
    np.testing.assert_allclose(xc1.sigma8(), 0.5, rtol=1e-9)


def test_xi_2d(cp, golden):
    g = golden('xi')
    sq, kq, zq = g['sq'], g['kq'], g['zq']
    interp = cp.Cosmology(engine='eisenstein_hu').get_fourier().pk_interpolator()
    xi2 = interp.to_xi()
    assert isinstance(xi2, cp.CorrelationFunctionInterpolator2D)
    np.testing.assert_allclose(xi2.s, g['xi2_s'], rtol=1e-13)
    np.testing.assert_allclose(xi2.z, g['xi2_z'], rtol=1e-14)
    assert tilted(xi2(sq, zq), g['xi2_eval'], sq) < 1e-11
    assert tilted(xi2(sq, zq, ignore_growth=True), g['xi2_eval_nogrowth'], sq) < 1e-11
    np.testing.assert_allclose(xi2(sq[:4], zq, grid=False), g['xi2_eval_pts'], rtol=1e-9)
    assert tilted(xi2.to_1d(z=0.35)(sq), g['xi2_to_1d_eval'], sq) < 1e-11
    assert np.isnan(xi2.sigma8_z(zq)).all()                     # default range: NaN, as in the reference
    c2 = interp.clone(extrap_kmin=1e-5, extrap_kmax=1e2)
    np.testing.assert_allclose(c2.pk, g['c2_pk'], rtol=1e-10)
    np.testing.assert_allclose(c2(kq, zq), g['c2_eval'], rtol=1e-9)
    xc2 = c2.to_xi()
    np.testing.assert_allclose(xc2.s, g['xc2_s'], rtol=1e-13)
    assert tilted(xc2(sq, zq), g['xc2_eval'], sq) < 1e-10
    pc2 = xc2.to_pk()
    assert isinstance(pc2, cp.PowerSpectrumInterpolator2D)
    np.testing.assert_allclose(pc2(kq, zq), g['xc2_to_pk_eval'], rtol=1e-9)
    np.testing.assert_allclose(xc2.sigma8_z(zq), g['xc2_sigma8_z'], rtol=1e-9)
    np.testing.assert_allclose(xc2.sigma_dz(zq), g['xc2_sigma_dz'], rtol=1e-9)
    assert xc2(sq[:3].astype('f4'), zq[:2].astype('f4')).dtype == np.float32 and xc2(1., 0.).shape == ()
    # from_callable mirrors (reference tests/test_interpolator.py:141-147)
    x2 = cp.CorrelationFunctionInterpolator2D.from_callable(xc2.s, xc2.z, xc2)
    s = np.logspace(-1, 2, 30)
    np.testing.assert_allclose(x2(s, zq), xc2(s, zq), rtol=1e-12)
    x1 = xc2.to_1d(z=0.)
    x1c = x1.from_callable(x1.s, x1)
    np.testing.assert_allclose(x1c(s), x1(s), rtol=1e-12)


def test_xi_tables(cp, golden):
    """Tabulated (s, z) and (s, columns) inputs against the reference goldens and the oracle restatement."""
    g = golden('xi')
    st, zt, tab, sq2 = g['tab_s'], g['tab_z'], g['tab_xi'], g['tab_sq']
    zq = np.array([0.1, 0.9, 1.7])
    t2 = cp.CorrelationFunctionInterpolator2D(st, zt, tab, interp_order_z=3)
    np.testing.assert_allclose(t2(sq2, zq), g['tab2_eval'], rtol=1e-9, atol=1e-14)
    np.testing.assert_allclose(t2(sq2, zq), oi.xi_interp_2d(st, zt, tab)(sq2, zq), rtol=1e-9, atol=1e-14)
    t1 = cp.CorrelationFunctionInterpolator1D(st, tab[:, :2])
    assert t1(sq2).shape == (sq2.size, 2)
    np.testing.assert_allclose(t1(sq2), g['tab1_eval'], rtol=1e-10, atol=1e-15)
    t1l = cp.CorrelationFunctionInterpolator1D(st, tab[:, 0], interp_s='lin')
    np.testing.assert_allclose(t1l(sq2), g['tab1_lin_eval'], rtol=1e-10, atol=1e-15)
    # unsorted input is sorted (reference interpolator.py:999-1005); clone() reproduces the interpolator
    perm = np.random.default_rng(0).permutation(st.size)
    np.testing.assert_allclose(cp.CorrelationFunctionInterpolator1D(st[perm], tab[perm, 0])(sq2), t1(sq2)[:, 0], rtol=1e-13, atol=1e-16)
    np.testing.assert_allclose(t2.clone()(sq2, zq), t2(sq2, zq), rtol=1e-13, atol=1e-16)
    with pytest.raises(ValueError):
        cp.CorrelationFunctionInterpolator2D(st, 0., tab[:, :1], interp_order_z=3)     # single-column table without growth_factor_sq (interpolator.py:1266-1267)


def test_kirkby2013(cp, golden):
    """Correlation-function BAO filter (reference bao_filter.py:835-909) on 1D / 2D xi, default and rescaled boxes."""
    from oracle import bao as obao
    g = golden('xi')
    sq, kq, zq = g['sq'], g['kq'], g['zq']
    fid = cp.Cosmology(engine='eisenstein_hu')
    interp = fid.get_fourier().pk_interpolator()
    xc1 = interp.to_1d(z=0.).clone(extrap_kmin=1e-5, extrap_kmax=1e2).to_xi()
    f1 = cp.CorrelationFunctionBAOFilter(xc1, engine='kirkby2013')
    np.testing.assert_allclose(f1.s, g['kirkby_s'], rtol=1e-13)
    np.testing.assert_allclose(f1.xi, g['kirkby1_xi'], rtol=1e-9, atol=1e-13)
    np.testing.assert_allclose(f1.xinow, g['kirkby1_xinow'], rtol=1e-9, atol=1e-13)
    np.testing.assert_allclose(f1.xinow, obao.kirkby2013(f1.s, f1.xi), rtol=1e-9, atol=1e-14)      # same input: operator == oracle
    np.testing.assert_allclose(f1.smooth_xi_interpolator()(sq), g['kirkby1_smooth_eval'], rtol=1e-9, atol=1e-13, equal_nan=True)
    other = cp.Cosmology(engine='eisenstein_hu', Omega_m=0.36, Omega_b=0.055, h=0.64, n_s=0.98, sigma8=0.85)
    f1r = cp.CorrelationFunctionBAOFilter(xc1, engine='kirkby2013', cosmo=other, cosmo_fid=fid)
    np.testing.assert_allclose(f1r.rs_drag_ratio(), g['kirkby1r_ratio'], rtol=1e-10)
    np.testing.assert_allclose(f1r.xinow, g['kirkby1r_xinow'], rtol=1e-9, atol=1e-13)
    f1d = cp.CorrelationFunctionBAOFilter(xc1, engine='kirkby2013', cosmo=other)
    np.testing.assert_allclose(f1d.rs_drag_ratio(), g['kirkby1d_ratio'], rtol=1e-10)
    np.testing.assert_allclose(f1d.xinow, g['kirkby1d_xinow'], rtol=1e-9, atol=1e-13)
    xc2 = interp.clone(extrap_kmin=1e-5, extrap_kmax=1e2).to_xi()
    f2 = cp.CorrelationFunctionBAOFilter(xc2, engine='kirkby2013', srange_left=(45., 80.), srange_right=(155., 195.), rescale_sbox=False, cosmo=other)
    assert f2.xinow.shape == (1024, xc2.z.size)
    np.testing.assert_allclose(f2.xinow[:, ::6], g['kirkby2_xinow'], rtol=1e-9, atol=1e-13)
    np.testing.assert_allclose(f2.smooth_xi_interpolator()(sq, zq), g['kirkby2_smooth_eval'], rtol=1e-9, atol=1e-13, equal_nan=True)
    assert np.isnan(f2.smooth_pk_interpolator()(kq, zq)).all() and np.isnan(g['kirkby2_smooth_pk_eval']).all()
    f1(xc1, cosmo=other)                      # re-run on new input (reference bao_filter.py:772-776)
    np.testing.assert_allclose(f1.xinow, g['kirkby1d_xinow'], rtol=1e-9, atol=1e-13)
    with pytest.raises(ValueError):
        cp.CorrelationFunctionBAOFilter(xc1, engine='nope')
