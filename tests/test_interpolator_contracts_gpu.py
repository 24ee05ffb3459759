"""GPU: the shape / dtype / ordering contracts the reference pins for its interpolators (tests/test_interpolator.py:8-32, 35-70,
168-300): scalar, empty and n-d inputs, f4 in -> f4 out, order independence at atol=0, clone equality, NaN / bounds_error."""
import warnings

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


QUERY = np.array([0.3, 0.15, 0.6, 0.2])


def assert_contract_1d(fn, trailing=()):
    """f(q): the result has the shape of q (then `trailing`), keeps a float32 query's width, and each value depends on its own query only."""
    for query in (0.25, np.empty(0), QUERY, QUERY.reshape(2, 2), np.tile(QUERY, (3, 1, 1))):
        assert np.shape(fn(query)) == np.shape(query) + trailing, np.shape(query)
    assert fn(QUERY.astype(np.float32)).dtype == np.float32 and fn(QUERY).dtype == np.float64
    order = np.argsort(QUERY)
    np.testing.assert_array_equal(fn(QUERY)[order], fn(QUERY[order]))


def assert_contract_2d(fn):
    """f(k, z): on a grid the result has shape(k) + shape(z); with grid=False k and z are paired and the result has their common shape; float32 in,
    float32 out; each entry depends on its own (k, z) only."""
    zq = np.array([0.4, 0.05, 0.2])
    shapes = [(), (0,), (4,), (2, 2), (3, 1, 4)]
    for sk in shapes:
        for sz in [(), (0,), (3,), (3, 1)]:
            kq = np.resize(QUERY, sk) if sk != (0,) else np.empty(0)
            zz = np.resize(zq, sz) if sz != (0,) else np.empty(0)
            assert np.shape(fn(kq, zz)) == sk + sz, (sk, sz)
    assert fn(QUERY.astype(np.float32), np.float32(0.2)).dtype == np.float32
    ok, oz = np.argsort(QUERY), np.argsort(zq)
    np.testing.assert_array_equal(fn(QUERY, zq)[np.ix_(ok, oz)], fn(QUERY[ok], zq[oz]))
    for sk in shapes:
        kq = np.resize(QUERY, sk) if sk != (0,) else np.empty(0)
        zz = np.resize(zq, sk) if sk != (0,) else np.empty(0)
        assert np.shape(fn(kq, zz, grid=False)) == sk, sk
    pairs = fn(QUERY[:3], zq, grid=False)
    np.testing.assert_allclose(pairs, np.diagonal(fn(QUERY[:3], zq)), rtol=1e-13, atol=0)       # pairs and grid take different kernels
    np.testing.assert_allclose(pairs[::-1], fn(QUERY[2::-1], zq[::-1], grid=False), rtol=1e-13, atol=0)


def test_power_spectrum_contracts():
    import torch
    assert torch.cuda.is_available()
    import cosmoprimo_amd as cp
    from cosmoprimo_amd import PowerSpectrumInterpolator1D, PowerSpectrumInterpolator2D
    warnings.simplefilter('ignore')
    cosmo = cp.Cosmology()
    k = np.geomspace(2e-3, 20., 90)
    pk = cp.Transfer(cosmo, engine='eisenstein_hu').transfer_k(k)**2 * k**cosmo['n_s']
    # one spectrum: the class, its clone, sigma(r)
    one = PowerSpectrumInterpolator1D(k, pk)
    assert_contract_1d(one)
    assert_contract_1d(one.sigma_r)
    twin = one.clone()
    probe = np.full((5, 3), 0.7)
    assert np.array_equal(twin(probe), one(probe)) and twin is not one
    # a table P(k, z) = P(k) for every z through the growth-factor form, and as an explicit (k, z) table
    znodes = np.array([0., 0.4, 0.9, 1.6])
    flat = PowerSpectrumInterpolator2D(k, z=0, pk=pk, growth_factor_sq=lambda z: np.ones_like(z))
    np.testing.assert_allclose(flat(k, z=np.linspace(0.1, 0.9, 7)), np.tile(pk[:, None], (1, 7)), rtol=1e-5, atol=0)
    table = PowerSpectrumInterpolator2D(k, z=znodes, pk=np.tile(pk[:, None], (1, znodes.size)))
    assert table(k, z=0.).shape == k.shape
    for interp in (flat, table):
        assert_contract_2d(interp)
    twin = flat.clone()
    assert np.array_equal(twin._pk, flat._pk)
    np.testing.assert_allclose(twin(k, z=[0., 0.]), flat(k, z=[0., 0.]), rtol=1e-18, atol=1e-18)
    # the callable an analytic engine hands out
    engine_pk = cp.Fourier(cosmo, engine='eisenstein_hu').pk_interpolator()
    assert_contract_2d(engine_pk)
    assert_contract_1d(engine_pk.to_1d(z=0.))
    assert engine_pk.sigma_rz(np.linspace(1., 10., 3), np.linspace(0., 1., 4)).shape == (3, 4)
    assert engine_pk.sigma_dz(np.linspace(0., 1., 4)).shape == (4,)


def test_extrapolation_and_nan():
    """reference tests/test_interpolator.py:168-300, 328-337"""
    import cosmoprimo_amd as cp
    from cosmoprimo_amd import PowerSpectrumInterpolator1D, PowerSpectrumInterpolator2D
    k = np.logspace(-3, 1., 50)
    pk = k**-1.5
    for kwargs in [dict(), dict(extrap_kmin=1e-5, extrap_kmax=1e3)]:
        interp = PowerSpectrumInterpolator1D(k, pk, **kwargs)
        kmin, kmax = kwargs.get('extrap_kmin', 1e-7), kwargs.get('extrap_kmax', 1e2)
        assert np.isnan(interp(kmin * 0.5)) and np.isnan(interp(kmax * 2.))
        assert np.isfinite(interp(np.array([kmin, kmax]))).all()
        assert np.allclose(interp(np.array([1e-4, 50.])), np.array([1e-4, 50.])**-1.5, rtol=1e-6)   # log-log extrapolation of a power law is exact
        with pytest.raises(ValueError):
            interp(kmax * 2., bounds_error=True)
    interp = PowerSpectrumInterpolator1D(k, pk, extrap_pk='lin')
    assert np.isnan(interp(5e-4)) and np.isfinite(interp(2e-3))
    with pytest.raises(ValueError):
        PowerSpectrumInterpolator1D(k, pk, interp_k='lin', extrap_pk='log')
    z = np.linspace(0., 1., 5)
    interp2 = PowerSpectrumInterpolator2D(k, z, pk[:, None] * (1 + z))
    assert np.isnan(interp2(1., 1.5)) and np.isnan(interp2(1e3, 0.5)) and np.isfinite(interp2(1., 0.5))
    with pytest.raises(ValueError):
        interp2(1., 1.5, bounds_error=True)
    bad = pk.copy()
    bad[10] *= -1
    assert np.isnan(PowerSpectrumInterpolator1D(k, bad)(k)).all()
    assert np.isnan(PowerSpectrumInterpolator2D(k, z, bad[:, None] * (1 + z))(k, z)).all()


def test_linear_order():
    """k = 1 (reference jax.py:176-177: scipy interp1d 'linear'): values, NaN outside, linear extrapolation, columns, log axes."""
    import cosmoprimo_amd.interpolator as it
    from scipy.interpolate import interp1d
    rng = np.random.default_rng(3)
    x = np.sort(rng.uniform(0.1, 10., 40))
    fun = rng.uniform(1., 2., (40, 3))
    xq = np.concatenate([[0.05, 20.], rng.uniform(0.1, 10., 30), x[[0, -1, 7]]])
    for extrap in (False, True):
        ref = interp1d(x, fun, kind='linear', axis=0, bounds_error=False, fill_value='extrapolate' if extrap else np.nan, assume_sorted=True)(xq)
        got = it.Interpolator1D(x, fun, k=1, extrap=extrap)(xq)
        np.testing.assert_allclose(got, ref, rtol=1e-13, equal_nan=True)
    xin = np.clip(xq[2:], x[0], x[-1])
    ref = 10**interp1d(np.log10(x), np.log10(fun[:, 0]), kind='linear')(np.log10(xin))
    np.testing.assert_allclose(it.Interpolator1D(x, fun[:, 0], k=1, interp_x='log', interp_fun='log')(xin), ref, rtol=1e-12)
    pk = it.PowerSpectrumInterpolator1D(x, fun[:, 0], interp_order_k=1)
    assert np.isfinite(pk(np.array([1e-6, 1., 50.]))).all()
    # k = 2: scipy interp1d 'quadratic' (make_interp_spline(k=2)), the other order the reference's Interpolator1D accepts
    for extrap in (False, True):
        ref = interp1d(x, fun, kind='quadratic', axis=0, bounds_error=False, fill_value='extrapolate' if extrap else np.nan, assume_sorted=True)(xq)
        got = it.Interpolator1D(x, fun, k=2, extrap=extrap)(xq)
        assert got.shape == ref.shape and np.array_equal(np.isnan(got), np.isnan(ref))
        np.testing.assert_allclose(got[np.isfinite(ref)], ref[np.isfinite(ref)], rtol=1e-10, atol=1e-12 * np.abs(fun).max())
    ref = 10**interp1d(np.log10(x), np.log10(fun[:, 0]), kind='quadratic')(np.log10(xin))
    np.testing.assert_allclose(it.Interpolator1D(x, fun[:, 0], k=2, interp_x='log', interp_fun='log')(xin), ref, rtol=1e-10)
    pk2 = it.PowerSpectrumInterpolator1D(x, fun[:, 0], interp_order_k=2)      # knots padded to (extrap_kmin, extrap_kmax) as in the reference
    knots, values = pk2._interp._x, pk2._interp._rows.cpu().numpy()[0]
    assert knots.size == x.size + 4 and np.allclose(knots[2:-2], np.log10(x))      # two log-extrapolated points on either side
    np.testing.assert_allclose(pk2(xin), 10**interp1d(knots, values, kind='quadratic')(np.log10(xin)), rtol=1e-10)
    with pytest.raises(NotImplementedError):
        it.Interpolator1D(x, fun, k=4)


def test_many_points_path():
    """Interpolator1D at more than 16 384 points takes the point-evaluation kernel (cp_spline_points) instead of a (queries x knots) operator:
    same numbers, queries that live on the device stay there; DistanceToRedshift for a catalogue (reference tests/test_utils.py::test_redshift_array)."""
    import torch
    import cosmoprimo_amd.interpolator as it
    from cosmoprimo_amd.utils import DistanceToRedshift
    from cosmoprimo_amd.fiducial import DESI
    from scipy.interpolate import CubicSpline
    rng = np.random.default_rng(9)
    x = np.sort(rng.uniform(0.1, 10., 300))
    fun = rng.uniform(1., 2., (300, 3))
    xq = np.concatenate([[0.05, 20.], rng.uniform(0.1, 10., 40000), x[[0, -1, 7]]])
    for extrap in (False, True):
        interp = it.Interpolator1D(x, fun, extrap=extrap)
        got = interp(xq)
        ref = CubicSpline(x, fun, axis=0, bc_type='natural', extrapolate=True)(xq)
        if not extrap:
            ref[(xq < x[0]) | (xq > x[-1])] = np.nan
        np.testing.assert_allclose(got, ref, rtol=1e-11, atol=1e-13, equal_nan=True)
        for nu in (1, 2):
            refd = CubicSpline(x, fun, axis=0, bc_type='natural', extrapolate=True)(xq, nu=nu)
            if not extrap:
                refd[(xq < x[0]) | (xq > x[-1])] = np.nan
            np.testing.assert_allclose(interp(xq, dx=nu), refd, rtol=1e-9, atol=1e-9, equal_nan=True)
        saved, it.Interpolator1D._npoints_operator = it.Interpolator1D._npoints_operator, 10**9      # the operator path on the same queries
        try:
            np.testing.assert_allclose(got, interp(xq), rtol=1e-12, atol=1e-12, equal_nan=True)
        finally:
            it.Interpolator1D._npoints_operator = saved
    loglog = it.Interpolator1D(x, fun[:, 0], interp_x='log', interp_fun='log')
    xin = np.clip(xq, x[0], x[-1])
    np.testing.assert_allclose(loglog(xin), 10**CubicSpline(np.log10(x), np.log10(fun[:, 0]), bc_type='natural')(np.log10(xin)), rtol=1e-11)
    with pytest.raises(ValueError):
        it.Interpolator1D(x, fun)(xq, bounds_error=True)
    cosmo = DESI()
    redshift = DistanceToRedshift(distance=cosmo.comoving_radial_distance, zmax=10., nz=4096)
    z = torch.rand(2000000, device='cuda:0', dtype=torch.float64) * 2.
    back = redshift(cosmo.comoving_radial_distance(z))
    assert back.is_cuda and back.shape == z.shape and float((back - z).abs().max()) < 1e-6
    zh = np.random.default_rng(1).uniform(0., 2., 10000)
    assert np.allclose(redshift(cosmo.comoving_radial_distance(zh)), zh, atol=1e-6)


def test_many_points_2d():
    """Interpolator2D / tabulated PowerSpectrumInterpolator2D at a mesh of wavenumbers (more than 16 384) and a few redshifts: the y direction by
    operator, the x direction point by point -- RectBivariateSpline's numbers, device-resident queries left on the device."""
    import torch
    import cosmoprimo_amd as cp
    import cosmoprimo_amd.interpolator as it
    from scipy.interpolate import RectBivariateSpline
    rng = np.random.default_rng(4)
    x, y = np.sort(rng.uniform(0., 5., 60)), np.linspace(0., 2., 12)
    fun = np.sin(x)[:, None] * np.cos(y)[None, :] + 0.1 * rng.standard_normal((60, 12))
    xq, yq = np.concatenate([rng.uniform(x[0], x[-1], 30000), [x[0], x[-1], -1., 7.]]), np.array([0., 0.33, 1.7, 2., 2.5])
    order = np.argsort(xq)
    ref = np.empty((xq.size, yq.size))
    ref[order] = RectBivariateSpline(x, y, fun, kx=3, ky=3, s=0)(xq[order], yq, grid=True)
    for extrap in (True, False):
        interp = it.Interpolator2D(x, y, fun, extrap=extrap)
        got = interp(xq, yq)
        expected = ref.copy()
        if not extrap:
            expected[(xq < x[0]) | (xq > x[-1])] = np.nan
            expected[:, (yq < y[0]) | (yq > y[-1])] = np.nan
        assert got.shape == (xq.size, yq.size) and np.array_equal(np.isnan(got), np.isnan(expected))
        keep = (xq >= x[0]) & (xq <= x[-1])        # (FITPACK clamps outside its knots; the reference masks there unless extrap)
        np.testing.assert_allclose(got[keep][:, :4], expected[keep][:, :4], rtol=1e-10, atol=1e-12)
        np.testing.assert_allclose(got[:20], interp(xq[:20], yq), rtol=1e-11, atol=1e-13, equal_nan=True)      # the operator path on a few of them
        np.testing.assert_allclose(got[-4:], interp(xq[-4:], yq), rtol=1e-11, atol=1e-13, equal_nan=True)      # ... and on the end knots and the points outside
        if extrap:      # outside the knots: FITPACK's value at the end knot, in x and in y
            np.testing.assert_allclose(got[~keep], expected[~keep], rtol=1e-10, atol=1e-12)
    k, z = np.geomspace(1e-4, 10., 200), np.linspace(0., 2., 8)
    table = cp.Cosmology(engine='eisenstein_hu').get_fourier().pk_interpolator()(k, z)
    pk2d = cp.PowerSpectrumInterpolator2D(k, z, table)
    kmesh = torch.as_tensor(10.**rng.uniform(-3.9, 0.9, 300000), device='cuda:0')
    out = pk2d(kmesh, np.array([0.5, 1.]))
    assert out.is_cuda and out.shape == (300000, 2)
    some = kmesh[:50].cpu().numpy()
    np.testing.assert_allclose(out[:50].cpu().numpy(), pk2d(some, np.array([0.5, 1.])), rtol=1e-10)


def test_many_pairs_2d_and_the_size_of_operators():
    """Pairs of (x, y) beyond one pass of the pair route come in pieces; an operator whose dense staging would not fit a host is refused, not built."""
    import cosmoprimo_amd.interpolator as it
    from cosmoprimo_amd.spline import LinearOperator, dense_operator
    rng = np.random.default_rng(8)
    x, y = np.sort(rng.uniform(0., 5., 40)), np.linspace(0., 2., 9)
    fun = np.sin(x)[:, None] * np.cos(y)[None, :]
    interp = it.Interpolator2D(x, y, fun)
    xq, yq = rng.uniform(-0.2, 5.2, 3500), rng.uniform(-0.1, 2.1, 3500)
    whole = interp(xq, yq, grid=False)
    try:
        it.Interpolator2D._pairs_chunk = 1000
        pieces = interp(xq, yq, grid=False)
        mesh = interp(xq.reshape(70, 50), yq.reshape(70, 50), grid=False)
    finally:
        it.Interpolator2D._pairs_chunk = 1 << 16
    assert pieces.shape == (3500,) and mesh.shape == (70, 50)
    assert np.array_equal(pieces, whole, equal_nan=True) and np.array_equal(mesh.ravel(), whole, equal_nan=True) and np.isnan(whole).any()
    # linear interpolation of a catalogue: the operator route in pieces of queries (200 000 x 400 weights would be 640 MB as one matrix)
    xk = np.sort(rng.uniform(0., 5., 400))
    yk = np.stack([np.sin(xk), np.cos(xk)], axis=-1)
    lin = it.Interpolator1D(xk, yk, k=1)
    xc = rng.uniform(-0.1, 5.1, 200000)
    got = lin(xc)
    inside = (xc >= xk[0]) & (xc <= xk[-1])
    assert got.shape == (200000, 2) and np.isnan(got[~inside]).all()
    np.testing.assert_allclose(got[inside, 0], np.interp(xc[inside], xk, yk[:, 0]), rtol=1e-12, atol=1e-14)
    np.testing.assert_allclose(got[inside, 1], np.interp(xc[inside], xk, yk[:, 1]), rtol=1e-12, atol=1e-14)
    assert lin(xc.reshape(400, 500)).shape == (400, 500, 2)
    with pytest.raises(MemoryError):
        it._linear_interp_operator(xk, np.linspace(0., 5., 1 << 21))
    knots = np.linspace(0., 1., 600)
    with pytest.raises(NotImplementedError):
        LinearOperator.spline(knots, np.linspace(0., 1., 1 << 20))      # 6.3e8 weights: 5 GB on the host
    with pytest.raises(MemoryError):
        dense_operator(knots, np.linspace(0., 1., 1 << 20))


def test_device_mesh_through_callable():
    """P(k, z) of an analytic engine at a mesh of wavenumbers that lives on the device: evaluated there (no copy of k to the host and back),
    same numbers as for the host array, NaN outside the extrapolation range."""
    import torch
    import cosmoprimo_amd as cp
    pk = cp.Cosmology(engine='eisenstein_hu').get_fourier().pk_interpolator()
    rng = np.random.default_rng(6)
    kh = np.concatenate([10.**rng.uniform(-6.5, 1.9, 100000), [1e-9, 1e3]])
    z = np.array([0., 1.])
    ref = pk(kh, z)
    out = pk(torch.as_tensor(kh, device='cuda:0'), z)
    assert out.is_cuda and out.shape == (kh.size, 2)
    assert np.array_equal(out.cpu().numpy(), ref, equal_nan=True) and np.isnan(ref[-2:]).all() and np.isfinite(ref[:-2]).all()
    mesh = torch.as_tensor(kh[:-2], device='cuda:0').reshape(100, 1000)
    assert pk(mesh, 0.5).shape == (100, 1000) and pk(mesh.to(torch.float32), 0.5).dtype == torch.float32
    with pytest.raises(ValueError):
        pk(torch.as_tensor(kh, device='cuda:0'), z, bounds_error=True)


def test_2d_spline_degrees():
    """Interpolator2D(kx, ky) and the interp_order_k / interp_order_z of the tabulated P(k, z) class for every degree RectBivariateSpline takes
    (reference jax.py:241-242, interpolator.py:667): grids, pairs, clamping beyond the data with extrap=True, NaN there without."""
    import cosmoprimo_amd as cp
    import cosmoprimo_amd.interpolator as it
    from scipy.interpolate import RectBivariateSpline
    rng = np.random.default_rng(7)
    x, y = np.sort(rng.uniform(0., 5., 40)), np.linspace(0., 2., 9)
    fun = np.sin(x)[:, None] * np.cos(y)[None, :] + 0.1 * rng.standard_normal((40, 9))
    xq, yq = np.sort(np.concatenate([rng.uniform(x[0], x[-1], 64), [-1., 6.]])), np.array([-0.5, 0., 0.33, 1.7, 2., 2.5])
    for kx, ky in [(1, 1), (2, 3), (3, 2), (4, 5), (5, 1), (3, 3)]:
        ref = RectBivariateSpline(x, y, fun, kx=kx, ky=ky, s=0)
        for extrap in (True, False):
            interp = it.Interpolator2D(x, y, fun, kx=kx, ky=ky, extrap=extrap)
            expected, pairs = ref(xq, yq, grid=True), ref(xq[:6], yq, grid=False)
            if not extrap:
                expected[(xq < x[0]) | (xq > x[-1])] = np.nan
                expected[:, (yq < y[0]) | (yq > y[-1])] = np.nan
                pairs[(xq[:6] < x[0]) | (yq < y[0]) | (yq > y[-1])] = np.nan
            np.testing.assert_allclose(interp(xq, yq), expected, rtol=1e-10, atol=1e-12, err_msg=str((kx, ky, extrap)))
            np.testing.assert_allclose(interp(xq[:6], yq, grid=False), pairs, rtol=1e-10, atol=1e-12, err_msg=str((kx, ky, extrap)))
    with pytest.raises(ValueError):
        it.Interpolator2D(x, y, fun, kx=6)
    # through the P(k, z) class: log10 k, log10 P, the padded wavenumbers
    k, z = np.geomspace(1e-3, 10., 120), np.linspace(0., 2., 8)
    table = cp.Cosmology(engine='eisenstein_hu').get_fourier().pk_interpolator()(k, z)
    kq, zq = np.geomspace(2e-3, 9., 50), np.array([0.1, 0.9, 1.9])
    for ok, oz in [(1, 1), (2, 2), (3, 1), (5, 4)]:
        got = cp.PowerSpectrumInterpolator2D(k, z, table, extrap_pk='lin', interp_order_k=ok, interp_order_z=oz)(kq, zq)
        ref = 1. * RectBivariateSpline(np.log10(k), z, table, kx=ok, ky=oz, s=0)(np.log10(kq), zq, grid=True)
        np.testing.assert_allclose(got, ref, rtol=1e-10, err_msg=str((ok, oz)))
        sig = cp.PowerSpectrumInterpolator2D(k, z, table, interp_order_k=ok, interp_order_z=oz).sigma8_z(zq)
        np.testing.assert_allclose(sig, cp.PowerSpectrumInterpolator2D(k, z, table).sigma8_z(zq), rtol=5e-3)


def test_pairs_kernel_against_numpy():
    """cp_bilinear_pairs: out[b, q] = sum_ij wx[q, i] f[b, i, j] wy[q, j] against numpy, odd sizes, more than 64 columns, zero and NaN weights."""
    import torch
    from cosmoprimo_amd import _lib, _device as dv
    dev = torch.device('cuda', 0)
    rng = np.random.default_rng(2)
    for nb, nq, nx, ny in ((1, 1, 4, 3), (3, 77, 50, 30), (2, 130, 33, 100)):
        wx, wy, f = rng.normal(size=(nq, nx)), rng.normal(size=(nq, ny)), rng.normal(size=(nb, nx, ny))
        wx[rng.uniform(size=wx.shape) < 0.7] = 0.
        if nq > 5:
            wx[5, 1] = np.nan
        twx, twy, tf = (torch.as_tensor(v, device=dev) for v in (wx, wy, f))
        out = torch.empty((nb, nq), dtype=torch.float64, device=dev)
        _lib.check(_lib.load().cp_bilinear_pairs(twx.data_ptr(), twy.data_ptr(), tf.data_ptr(), out.data_ptr(), nb, nq, nx, ny, 0, dv.stream_of(dev)))
        ref = np.einsum('qi,bij,qj->bq', wx, f, wy)
        got = out.cpu().numpy()
        assert np.array_equal(np.isnan(got), np.isnan(ref))
        np.testing.assert_allclose(got[np.isfinite(ref)], ref[np.isfinite(ref)], rtol=0, atol=1e-13 * np.abs(ref[np.isfinite(ref)]).max())
