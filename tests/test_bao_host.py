"""CPU: host-side pieces of the BAO filters against scipy (what the reference calls): quadratic interp1d operator, peak search."""
import numpy as np
from scipy import interpolate, signal

from cosmoprimo_amd.bao_filter import _quadratic_interp_operator, _local_maxima, RegisteredPowerSpectrumBAOFilter


def test_quadratic_interp_operator():
    rng = np.random.default_rng(0)
    x = np.linspace(0., 1., 341)**1.3
    for ix in [np.array([0, 161, 215, 244, 262, 275, 285, 294, 302, 308, 314, 339, 340]), np.array([0, 190, 232, 254, 269, 280, 290, 298, 304, 310, 315, 340])]:
        y = rng.normal(size=(ix.size, 3))
        ref = interpolate.interp1d(x[ix], y, kind=2, axis=0, fill_value='extrapolate', assume_sorted=True)(x)
        W = _quadratic_interp_operator(x[ix], x)
        assert np.abs(W.dot(y) - ref).max() < 1e-11 * np.abs(ref).max()


def test_local_maxima():
    rng = np.random.default_rng(1)
    for _ in range(20):
        x = np.round(rng.normal(size=200), 1)      # rounding creates plateaus
        assert np.array_equal(_local_maxima(x), signal.find_peaks(x)[0])


def test_registry():
    assert {'wallish2018', 'brieden2022'} <= set(RegisteredPowerSpectrumBAOFilter._registry)


def test_fitpack_interp_operator():
    """Interpolating splines of degree 1...5 with FITPACK's knots (RectBivariateSpline(s=0) along one axis), clamped beyond the data."""
    from cosmoprimo_amd.interpolator import _fitpack_interp_operator
    rng = np.random.default_rng(2)
    x = np.sort(rng.uniform(0., 4., 23))
    fun = rng.normal(size=(23, 7))
    xq = np.sort(np.concatenate([rng.uniform(x[0], x[-1], 50), x[:3], [x[0] - 0.5, x[-1] + 1.]]))
    yk = np.arange(7.)
    for k in range(1, 6):
        ref = interpolate.RectBivariateSpline(x, yk, fun, kx=k, ky=1, s=0)(xq, yk, grid=True) if k > 1 else None
        W = _fitpack_interp_operator(x, xq, k)
        if ref is None:
            ref = np.array([np.interp(xq, x, fun[:, j]) for j in range(7)]).T
        assert np.abs(W.dot(fun) - ref).max() < 1e-12 * np.abs(fun).max(), k
