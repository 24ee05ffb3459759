"""Oracle: numpy restatement of cosmoprimo's sigma(r) / sigma_d integrals (TEST INFRASTRUCTURE ONLY).

Follows /root/reference/cosmoprimo/interpolator.py: kernel_tophat2 (:90-120), integrate_sigma_d2 (:123-197),
integrate_sigma_r2 (:200-292; methods 'fftlog' (default) and 'simpson'), and cosmoprimo/jax.py simpson (:365-507, the
scipy.integrate.simpson composite rule) and Interpolator1D (natural CubicSpline, :169-175).

Parity status: PINNED by tests/golden/sigma.npz (G4).
"""
import numpy as np
from scipy.interpolate import CubicSpline

from . import fftlog as ofl


def kernel_tophat2(x):
    """W(x)^2 with the 5-term Maclaurin series below x = 0.1 (interpolator.py:90-120)."""
    x = np.asarray(x, dtype='f8')
    x2 = x**2
    low = 1. + x2 * (-1.0 / 10.0 + x2 * (1.0 / 280.0 + x2 * (-1.0 / 15120.0 + x2 * (1.0 / 1330560.0 + x2 * (-1.0 / 172972800.0)))))
    with np.errstate(all='ignore'):
        high = 3. * (np.sin(x) - x * np.cos(x)) / x**3
    return np.where(x < 0.1, low, high)**2


def _simpson_pairs(y, x):
    """Composite Simpson over consecutive interval pairs of an odd number of samples (unequal spacing), axis 0."""
    h = np.diff(x)
    h0, h1 = h[0::2], h[1::2]
    hsum, hprod, ratio = h0 + h1, h0 * h1, h0 / h1
    sh = (-1,) + (1,) * (y.ndim - 1)
    tmp = hsum.reshape(sh) / 6.0 * (y[0:-2:2] * (2 - 1.0 / ratio).reshape(sh) + y[1:-1:2] * (hsum * hsum / hprod).reshape(sh)
                                   + y[2::2] * (2 - ratio).reshape(sh))
    return np.sum(tmp, axis=0)


def simpson(y, x):
    """The reference's simpson (cosmoprimo/jax.py:365-507 = scipy v1.0.0, even='avg') along axis 0: for an even number of
    samples, the average of {Simpson on the first N-1 samples + trapezoid on the last interval} and {trapezoid on the first
    interval + Simpson on the last N-1 samples}."""
    y = np.asarray(y, dtype='f8')
    n = y.shape[0]
    if n % 2 == 1:
        return _simpson_pairs(y, x)
    val = 0.5 * (x[-1] - x[-2]) * (y[-1] + y[-2]) + 0.5 * (x[1] - x[0]) * (y[1] + y[0])
    result = _simpson_pairs(y[:-1], x[:-1]) + _simpson_pairs(y[1:], x[1:])
    return val / 2.0 + result / 2.0


def sigma_r2(r, pk, kmin=1e-7, kmax=1e2, method='fftlog', nk=1024):
    """
    integrate_sigma_r2 (interpolator.py:200-292).  ``pk`` : callable k -> (nk,) or (nk, ncol).  Returns shape r.shape (+ (ncol,)).
    """
    r = np.asarray(r, dtype='f8')
    rshape = r.shape
    rr = r.ravel()
    pshape = np.shape(pk(np.array([kmin])))[1:]
    if method == 'fftlog':
        k = np.geomspace(kmin, kmax, nk)                                             # :287
        p = pk(k).reshape(k.shape + (-1,))
        t = ofl.tophat_variance(k)
        var = ofl.apply(t, p.T[:, None, :])[:, 0]                                     # (ncol, nk)
        s = t.y[0]
        tmp = (2. * np.pi**2) * CubicSpline(s, var.T, axis=0, bc_type='natural', extrapolate=False)(rr)   # :289
        tmp = np.where(((rr >= s[0]) & (rr <= s[-1]))[:, None], tmp, np.nan)
    elif method == 'simpson':
        limits = (np.log(kmin * (1. + 1e-9)), np.log(kmax * (1. - 1e-9)))           # :251
        logk = np.linspace(*limits, nk)
        k = np.exp(logk)
        p = pk(k).reshape(k.shape + (-1,))
        y = kernel_tophat2(k[:, None] * rr)[:, :, None] * (k[:, None]**3 * p)[:, None, :]   # :244
        tmp = simpson(y, logk)
    elif method == 'leggauss':                                                      # :274-280 ("not accurate")
        limits = (np.log(kmin * (1. + 1e-9)), np.log(kmax * (1. - 1e-9)))
        x, wx = np.polynomial.legendre.leggauss(100 if nk == 1024 else nk)
        logk = (limits[1] - limits[0]) / 2. * (1. + x) + limits[0]
        k = np.exp(logk)
        p = pk(k).reshape(k.shape + (-1,))
        y = kernel_tophat2(k[:, None] * rr)[:, :, None] * (k[:, None]**3 * p)[:, None, :]
        tmp = np.sum(y * ((limits[1] - limits[0]) / 2. * wx)[:, None, None], axis=0)
    else:
        raise ValueError(method)
    return (1. / (2. * np.pi**2) * tmp).reshape(rshape + pshape)                    # :290-291


def sigma_d2(pk, kmin=1e-7, kmax=1e2, nk=1024):
    """integrate_sigma_d2, method='simpson' (interpolator.py:190-196): 1/(6 pi^2) int dk P(k)."""
    limits = (np.log(kmin * (1. + 1e-9)), np.log(kmax * (1. - 1e-9)))
    logk = np.linspace(*limits, nk)
    k = np.exp(logk)
    p = pk(k)
    y = (k * p.T).T
    return 1. / (6. * np.pi**2) * simpson(y, logk)
