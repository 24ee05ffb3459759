"""
ORACLE (test infrastructure, CPU, numpy / scipy) -- not part of the shipped path.

Restatement of the reference's tabulated interpolators and of the P(k) <-> xi(s) conversions built on them:
  cosmoprimo/jax.py:135-209            Interpolator1D  (k = 3, numpy branch: natural CubicSpline along axis 0)
  cosmoprimo/jax.py:213-287            Interpolator2D  (RectBivariateSpline, s = 0)
  cosmoprimo/interpolator.py:42-87     _pad_log
  cosmoprimo/interpolator.py:329-351   _BasePowerSpectrumInterpolator._prepare
  cosmoprimo/interpolator.py:419-521   PowerSpectrumInterpolator1D (constructor, __call__)
  cosmoprimo/interpolator.py:616-817   PowerSpectrumInterpolator2D (constructor, __call__)
  cosmoprimo/interpolator.py:584-605, 965-987    to_xi
  cosmoprimo/interpolator.py:994-1222, 1225-1498 CorrelationFunctionInterpolator1D / 2D, to_pk
Only the behaviour on float64 numpy input is restated (the reference's jax branch does not run in this image).
Pinned by tests/golden/sigma.npz (table_* keys) and tests/golden/xi.npz, both generated from the imported reference
(oracle/gen_golden.py: gen_sigma, gen_xi).
"""
import numpy as np
from scipy.interpolate import CubicSpline, RectBivariateSpline

from . import fftlog as ofl


def pad_log(k, pk, extrap_kmin=1e-7, extrap_kmax=1e2):
    """Two log-log linearly extrapolated points on either side (interpolator.py:42-87).  Returns (log10 k, log10 pk)."""
    with np.errstate(invalid="ignore", divide="ignore"):   # a non-positive P gives NaN, which the spline wrapper propagates
        lk, lp = np.log10(k), np.log10(pk)
    lo = np.log10(min(extrap_kmin, k[0] * (1. - 1e-9)))
    hi = np.log10(max(extrap_kmax, k[-1] * (1. + 1e-9)))
    slope_hi = (lp[-1] - lp[-2]) / (lk[-1] - lk[-2])
    khi = np.array([0.1 * lk[-1] + 0.9 * hi, hi])
    phi = np.stack([lp[-1] + slope_hi * (kk - lk[-1]) for kk in khi])
    slope_lo = (lp[1] - lp[0]) / (lk[1] - lk[0])
    klo = np.array([lo, 0.1 * lk[0] + 0.9 * lo])
    plo = np.stack([lp[0] + slope_lo * (kk - lk[0]) for kk in klo])
    return np.concatenate([klo, lk, khi]), np.concatenate([plo, lp, phi], axis=0)


def spline1d(x, fun, interp_x='lin', interp_fun='lin', extrap=False):
    """Interpolator1D, k = 3 (jax.py:135-209): natural cubic spline along axis 0 of ``fun`` (n, ...); NaN outside [x0, x1]."""
    x = np.asarray(x, dtype='f8')
    fun = np.asarray(fun, dtype='f8')
    ix = np.argsort(x)
    x, fun = x[ix], fun[ix]
    shape = fun.shape[1:]
    xs = np.log10(x) if interp_x == 'log' else x
    fs = (np.log10(fun) if interp_fun == 'log' else fun).reshape(x.size, -1)
    good = ~np.isnan(fs).all(axis=0)            # all-NaN columns are set aside (:161-163)
    spl = None
    if good.any() and not np.isnan(fs[:, good]).any():   # any other NaN: everything NaN, no exception (:165-172)
        spl = CubicSpline(xs, fs[:, good], axis=0, bc_type='natural', extrapolate=bool(extrap))

    def call(xq, dx=0):
        xq = np.asarray(xq, dtype='f8')
        qs = xq.shape
        xq = xq.ravel()
        with np.errstate(all='ignore'):
            xx = np.log10(xq) if interp_x == 'log' else xq
        out = np.full((xq.size, fs.shape[1]), np.nan)
        if spl is not None:
            tmp = spl(xx, nu=dx)
            if interp_fun == 'log':
                tmp = 10**tmp
            if not extrap:
                tmp = np.where(((xq >= x[0]) & (xq <= x[-1]))[:, None], tmp, np.nan)
            out[:, good] = tmp
        return out.reshape(qs + shape)

    return call


def spline2d(x, y, fun, interp_x='lin', interp_fun='lin', extrap=False):
    """Interpolator2D, kx = ky = 3 (jax.py:213-287): RectBivariateSpline(s=0) on (x or log10 x, y); NaN outside the table."""
    x, y = np.asarray(x, dtype='f8'), np.asarray(y, dtype='f8')
    fun = np.asarray(fun, dtype='f8')
    ix, iy = np.argsort(x), np.argsort(y)
    x, y, fun = x[ix], y[iy], fun[np.ix_(ix, iy)]
    spl = RectBivariateSpline(np.log10(x) if interp_x == 'log' else x, y, np.log10(fun) if interp_fun == 'log' else fun, kx=3, ky=3, s=0)

    def call(xq, yq, grid=True):
        xq, yq = np.asarray(xq, dtype='f8'), np.asarray(yq, dtype='f8')
        shape = xq.shape + yq.shape if grid else xq.shape
        xq, yq = xq.ravel(), yq.ravel()
        mx, my = (xq >= x[0]) & (xq <= x[-1]), (yq >= y[0]) & (yq <= y[-1])
        with np.errstate(all='ignore'):
            xx = np.log10(xq) if interp_x == 'log' else xq
        if grid:
            jx, jy = np.argsort(xx), np.argsort(yq)     # FITPACK wants sorted grid queries (:266-269)
            tmp = spl(xx[jx], yq[jy], grid=True)[np.ix_(np.argsort(jx), np.argsort(jy))]
            mask = mx[:, None] & my
        else:
            tmp = spl(xx, yq, grid=False)
            mask = mx & my
        if interp_fun == 'log':
            tmp = 10**tmp
        if not extrap:
            tmp = np.where(mask, tmp, np.nan)
        return tmp.reshape(shape)

    return call


def pk_interp_1d(k, pk, extrap_kmin=1e-7, extrap_kmax=1e2):
    """PowerSpectrumInterpolator1D(k, pk) with the default log-log settings: callable k -> P (NaN outside the extrapolation range)."""
    k = np.asarray(k, dtype='f8').ravel()
    pk = np.asarray(pk, dtype='f8').reshape(k.shape + np.shape(pk)[1:])
    ix = np.argsort(k)
    lk, lp = pad_log(k[ix], pk[ix], extrap_kmin, extrap_kmax)        # :345-350
    return spline1d(10**lk, 10**lp, interp_x='log', interp_fun='log')


def pk_interp_2d(k, z, pk, extrap_kmin=1e-7, extrap_kmax=1e2, growth_factor_sq=None):
    """PowerSpectrumInterpolator2D with default settings: callable (k, z, grid=True, ignore_growth=False)."""
    k, z = np.asarray(k, dtype='f8').ravel(), np.asarray(z, dtype='f8').ravel()
    pk = np.asarray(pk, dtype='f8').reshape(k.size, -1)
    ik, iz = np.argsort(k), np.argsort(z)
    k, z, pk = k[ik], z[iz], pk[ik][:, iz] if pk.shape[1] > 1 else pk[ik]
    lk, lp = pad_log(k, pk, extrap_kmin, extrap_kmax)
    if pk.shape[1] > 1:
        base = spline2d(10**lk, z, 10**lp, interp_x='log', interp_fun='log')
    else:
        one = spline1d(10**lk, 10**lp[:, 0], interp_x='log', interp_fun='log')

        def base(kq, zq, grid=True):                                 # :800-811: the single column repeated along z
            tmp = one(kq)
            return np.repeat(tmp[..., None], np.size(zq), axis=-1).reshape(np.shape(kq) + np.shape(zq)) if grid else tmp

    def call(kq, zq, grid=True, ignore_growth=False):
        tmp = base(kq, zq, grid=grid)
        if growth_factor_sq is not None and not ignore_growth:
            tmp = tmp * growth_factor_sq(np.asarray(zq, dtype='f8'))
        return tmp

    return call


def xi_interp_1d(s, xi, interp_s='log'):
    """CorrelationFunctionInterpolator1D(s, xi) (interpolator.py:1077-1100): lin-y natural spline in log10 s, no extrapolation."""
    return spline1d(s, xi, interp_x=interp_s)


def xi_interp_2d(s, z, xi, interp_s='log', growth_factor_sq=None):
    """CorrelationFunctionInterpolator2D (interpolator.py:1229-1281, 1356-1407)."""
    xi = np.asarray(xi, dtype='f8').reshape(np.size(s), -1)
    if xi.shape[1] > 1:
        base = spline2d(s, z, xi, interp_x=interp_s)
    else:
        one = spline1d(s, xi[:, 0], interp_x=interp_s)

        def base(sq, zq, grid=True):
            tmp = one(sq)
            return np.repeat(tmp[..., None], np.size(zq), axis=-1).reshape(np.shape(sq) + np.shape(zq)) if grid else tmp

    def call(sq, zq, grid=True, ignore_growth=False):
        tmp = base(sq, zq, grid=grid)
        if growth_factor_sq is not None and not ignore_growth:
            tmp = tmp * growth_factor_sq(np.asarray(zq, dtype='f8'))
        return tmp

    return call


def to_xi(pk, kmin=1e-7, kmax=1e2, nk=1024):
    """to_xi (interpolator.py:584-605, 965-987): ``pk`` : callable k -> (nk,) or (nk, ncol).  Returns (s, xi (nk[, ncol]))."""
    k = np.geomspace(kmin, kmax, nk)
    p = pk(k)
    t = ofl.power_to_correlation(k)
    rows = p.reshape(nk, -1).T[:, None, :]
    xi = ofl.apply(t, rows)[:, 0]
    return t.y[0], xi.T.reshape(p.shape)


def to_pk(xi, smin, smax, ns=1024):
    """to_pk (interpolator.py:1201-1222, 1477-1498): ``xi`` : callable s -> (ns,) or (ns, ncol).  Returns (k, pk (ns[, ncol]))."""
    s = np.geomspace(smin, smax, ns)
    x = xi(s)
    t = ofl.correlation_to_power(s)
    rows = x.reshape(ns, -1).T[:, None, :]
    pk = ofl.apply(t, rows)[:, 0]
    return t.y[0], pk.T.reshape(x.shape)
