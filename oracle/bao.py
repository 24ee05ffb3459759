"""Oracle: numpy/scipy restatement of cosmoprimo's wallish2018 and brieden2022 BAO filters (TEST INFRASTRUCTURE ONLY).

Follows /root/reference/cosmoprimo/bao_filter.py: base class set_k / set_pk (:81-102), Wallish2018 _compute / _tophat
(:361-431), Brieden2022 _prepare / _interp / _compute (:461-509), and utils.LeastSquareSolver (utils.py:144-272).
Inputs are callables pk(k) -> (nk, ncol) so that the oracle is independent of the interpolator classes.

The remaining P(k) filters of the registry (hinton2017 :172-241, savgol :244-266, ehsavgol :269-286,
ehpoly :289-342, peakaverage :512-580) are restated array-in / array-out further down.

Parity status: PINNED by tests/golden/bao.npz (G6) and tests/golden/bao2.npz (f2).
"""
import numpy as np
from scipy import fftpack, interpolate, signal


def filter_k(extrap_kmin=1e-7, extrap_kmax=1e2, nk=1024):
    return np.geomspace(extrap_kmin, extrap_kmax, nk)      # bao_filter.py:81-90


def _tophat(k, kmax=1, scale=1):
    tophat = np.ones_like(k)
    mask = k > kmax
    tophat[mask] *= np.exp(-scale**2 * (k[mask] / kmax - 1.)**2)   # :426-431
    return tophat


def wallish2018(pk, extrap_kmin=1e-7, extrap_kmax=1e2, nk=1024, return_intermediates=False):
    """pk : callable k -> (len(k), ncol).  Returns pknow (nk, ncol) on filter_k (bao_filter.py:361-424)."""
    kf = filter_k(extrap_kmin, extrap_kmax, nk)
    pkf = pk(kf)
    k = np.linspace(extrap_kmin, 2., 4096)
    p = pk(k)
    kpk = np.log(k[:, None] * p)
    kpkffted = fftpack.dst(kpk, type=2, axis=0, norm='ortho')
    even, odd = kpkffted[::2].copy(), kpkffted[1::2].copy()
    xeven, xodd = 1 + np.arange(even.shape[0]), 1 + np.arange(odd.shape[0])
    dd_even = interpolate.CubicSpline(xeven, even, axis=0, bc_type='clamped', extrapolate=False)(xeven, nu=2)
    dd_odd = interpolate.CubicSpline(xodd, odd, axis=0, bc_type='clamped', extrapolate=False)(xodd, nu=2)
    margin_first, margin_second, offset = 20, 5, (-10, 20)
    boxes = []

    def smooth(y, dd, x):
        argmax = dd[margin_first:-margin_first].argmax() + margin_first
        ibox = (argmax + offset[0], argmax + margin_second + dd[argmax + margin_second:-margin_first].argmax() + offset[1])
        mask = np.ones_like(y, dtype=np.bool_)
        mask[ibox[0]:ibox[1] + 1] = False
        boxes.append(ibox)
        spline = interpolate.CubicSpline(x[mask], y[mask] * x[mask]**2, axis=-1, bc_type='clamped', extrapolate=False)
        return spline(x) / x**2

    for iz in range(p.shape[-1]):
        even[:, iz] = smooth(even[:, iz], dd_even[:, iz], xeven)
        odd[:, iz] = smooth(odd[:, iz], dd_odd[:, iz], xodd)
    merged = np.empty_like(kpkffted)
    merged[::2], merged[1::2] = even, odd
    kpknow = fftpack.idst(merged, type=2, axis=0, norm='ortho')
    pknow = np.exp(kpknow) / k[..., None]
    mask = (k > 1e-2) & (k < 1.5)
    mask_left, mask_right = kf < 5e-4, kf > 2.
    kk = np.concatenate([kf[mask_left], k[mask], kf[mask_right]], axis=0)
    pp = np.concatenate([pkf[mask_left], pknow[mask], pkf[mask_right]], axis=0)
    pknow = interpolate.CubicSpline(kk, pp, axis=0, bc_type='clamped', extrapolate=False)(kf)
    wiggles = (pkf / pknow - 1.) * _tophat(kf, kmax=1., scale=20.)[..., None] + 1.
    out = pkf / wiggles
    if return_intermediates:
        return out, dict(dd_even=dd_even, dd_odd=dd_odd, even_now=even, odd_now=odd, boxes=boxes, kpkffted=kpkffted)
    return out


def least_squares_constrained(gradient, precision, constraint_gradient, delta, constraint):
    """LeastSquareSolver(gradient, precision (1D), constraint_gradient, compute_inverse=False)(delta, constraint) -> model (utils.py:161-272)."""
    hv = gradient * precision
    invfisher = hv.dot(gradient.T)
    nc = constraint_gradient.shape[-1]
    invfisher = np.block([[invfisher, -constraint_gradient], [constraint_gradient.T, np.zeros((nc, nc))]])
    hv = np.block([[hv, np.zeros(constraint_gradient.shape)], [np.zeros((nc, gradient.shape[-1])), np.eye(nc)]])
    d = np.concatenate([delta, np.atleast_1d(constraint)], axis=-1)
    params = np.linalg.solve(invfisher, hv.dot(d.T)).T[..., :gradient.shape[0]]
    return params.dot(gradient)


def _interp_envelopes(ixh, ixl, x, y, kind=2):
    toret = 0.
    for ix in [ixh, ixl]:
        toret = toret + interpolate.interp1d(x[ix], y[ix], kind=kind, axis=0, fill_value='extrapolate', assume_sorted=True)(x)   # :483-488
    return toret / 2.


def brieden2022_prepare(pk_fid, pknow_fid, extrap_kmin=1e-7, extrap_kmax=1e2, nk=1024):
    """pk_fid, pknow_fid : callables k -> (nk,) of the fiducial cosmology (its engine / its EH no-wiggle engine), z = 0 (bao_filter.py:461-480)."""
    kf = filter_k(extrap_kmin, extrap_kmax, nk)
    kmask_fid = (kf >= 1e-3) & (kf <= 1.)
    k_fid = kf[kmask_fid]
    ratio = pk_fid(k_fid) / pknow_fid(k_fid)
    gradient = np.array([k_fid**(i - 1) for i in range(4)])
    cg = np.column_stack([gradient[..., 0], gradient[..., 1] - gradient[..., 0], gradient[..., -1], gradient[..., -2] - gradient[..., -1]])
    model = least_squares_constrained(gradient, k_fid**2, cg, ratio, [ratio[..., 0], ratio[..., 1] - ratio[..., 0], ratio[..., -1], ratio[..., -2] - ratio[..., -1]])
    pknow_correction = model[:, None]
    ratio_fid = ratio[:, None] / pknow_correction
    ik0 = np.searchsorted(k_fid, 0.02, side='right') + 1
    peaks = []
    for si in [1., -1.]:
        ix = signal.find_peaks(si * ratio_fid[ik0:, 0])[0] + ik0
        ix = np.concatenate([[0]] * int(ix[0] > 0) + [ix] + [[-1]] * int(ix[-1] < k_fid.size - 1), axis=0)
        peaks.append(ix)
    ratio_now_fid = _interp_envelopes(*peaks, k_fid, ratio_fid)
    return dict(kmask_fid=kmask_fid, k_fid=k_fid, pknow_correction=pknow_correction, ratio_fid=ratio_fid, peaks=peaks, ratio_now_fid=ratio_now_fid)


def brieden2022_compute(prep, pk, pknow_cosmo, rescale, clone_eval, extrap_kmin=1e-7, extrap_kmax=1e2, nk=1024):
    """
    pk : callable k -> (nk, ncol) input spectrum; pknow_cosmo : callable k -> (nk,) EH no-wiggle P(k, z=0) of `cosmo`;
    rescale = rs_drag(cosmo) / rs_drag(cosmo_fid); clone_eval(k_knots, pk_knots, k_eval) -> the input interpolator cloned on
    (k_knots, pk_knots) and evaluated at k_eval (bao_filter.py:490-509).
    """
    kf = filter_k(extrap_kmin, extrap_kmax, nk)
    k_fid = prep['k_fid']
    p = pk(k_fid / rescale)
    p = p.reshape(p.shape[0], -1)
    pknow = pknow_cosmo(k_fid * rescale)[:, None] * prep['pknow_correction']
    ratio = p / pknow / prep['ratio_fid']
    pknow = _interp_envelopes(*prep['peaks'], k_fid, ratio) * pknow * prep['ratio_now_fid']
    out = pk(kf).reshape(kf.size, -1).copy()
    out[prep['kmask_fid']] = clone_eval(k_fid / rescale, pknow, k_fid).reshape(k_fid.size, -1)
    return out


def kirkby2013(s, xi, rescale=1., srange_left=(50., 82.), srange_right=(150., 190.)):
    """
    Kirkby2013CorrelationFunctionBAOFilter._prepare / _compute (bao_filter.py:883-909): weighted fit of a_0 s + a_1 + a_2 / s +
    a_3 / s^2 + a_4 / s^3 on the two side bands, blended into xi between them.  ``xi`` : (ns,) or (ns, ncol); ``rescale`` =
    rs_drag ratio (:147-158, 896-898).  Returns xinow of the same shape.
    """
    s = np.asarray(s, dtype='f8')
    xi = np.asarray(xi, dtype='f8')
    shape = xi.shape
    xi = xi.reshape(s.size, -1)
    l, r = np.asarray(srange_left, dtype='f8'), np.asarray(srange_right, dtype='f8')
    smask = (s >= l[0] / 2.) & (s <= r[1] * 2.)                                   # safety factor 2 (:884-885)
    model = np.array([s**(1 - i) for i in range(5)])
    frac = 1. / 100.
    shift = (r[0] - l[1]) * frac
    wx = np.concatenate([[l[0] * (1. - frac)], l, [l[1] + shift, r[0] - shift], r, [r[1] * (1. + frac)]])
    wy = np.array([0., 1., 1., 0., 0., 1., 1., 0.])
    precision = np.interp(s[smask] / rescale, wx, wy, left=0., right=0.)          # :900
    center = np.interp(s / rescale, wx[2:-2], 1. - wy[2:-2], left=0., right=0.)   # :902
    g = model[:, smask]
    hv = g * precision                                                            # LeastSquareSolver without constraints (utils.py:161-272)
    params = np.linalg.solve(hv.dot(g.T), hv.dot(xi[smask])).T                    # (ncol, 5)
    fit = params.dot(model)                                                       # (ncol, ns)
    return (xi.T * (1. - center) + fit * center).T.reshape(shape)


def least_squares_projector(gradient, precision, constraint_gradient):
    """LeastSquareSolver(..., compute_inverse=True).projector (utils.py:161-232): explicit inverse, as the reference computes it."""
    hv = gradient * precision
    invfisher = hv.dot(gradient.T)
    nc = constraint_gradient.shape[-1]
    invfisher = np.block([[invfisher, -constraint_gradient], [constraint_gradient.T, np.zeros((nc, nc))]])
    hv = np.block([[hv, np.zeros(constraint_gradient.shape)], [np.zeros((nc, gradient.shape[-1])), np.eye(nc)]])
    return np.linalg.inv(invfisher).dot(hv).T


def hinton2017(k, pk, degree=12, sigma=0.5, weight=0.9):
    """Hinton2017 (bao_filter.py:215-241): constrained degree-12 polynomial in log10 k fitted to log10 P on 1e-4 < k < 5.
    ``pk`` (nk, ncol); the Gaussian down-weighting is centred on the maximum of the FIRST column (:219)."""
    pk = pk.reshape(k.size, -1)
    kmask = (k > 1e-4) & (k < 5.)
    logk = np.log10(k[kmask])
    logpk = np.log10(pk[kmask].T)
    maxk = logk[np.argmax(logpk[0], axis=0)]
    w = 1. - weight * np.exp(-0.5 * ((logk - maxk) / sigma)**2)
    gradient = np.array([((logk - np.mean(logk)) / np.std(logk))**i for i in range(degree + 1)])
    cg = np.column_stack([gradient[..., 0], gradient[..., 1] - gradient[..., 0], gradient[..., 2] - 2. * gradient[..., 1] + gradient[..., 0],
                          gradient[..., -1], gradient[..., -2] - gradient[..., -1], gradient[..., -3] - 2. * gradient[..., -2] + gradient[..., -1]])
    constraint = np.column_stack([logpk[..., 0], logpk[..., 1] - logpk[..., 0], logpk[..., 2] - 2. * logpk[..., 1] + logpk[..., 0],
                                  logpk[..., -1], logpk[..., -2] - logpk[..., -1], logpk[..., -3] - 2. * logpk[..., -2] + logpk[..., -1]])
    params = np.concatenate([logpk, constraint], axis=-1).dot(least_squares_projector(gradient, w**2, cg))[..., :degree + 1]
    out = pk.copy()
    out[kmask] = 10**params.dot(gradient).T
    return out


def savgol_length(k):
    return int(np.ceil(np.log(7) / np.log(k[-1] / k[-2])) // 2 * 2 + 1)     # :262, 283


def savgol(k, pk):
    """SavGol (bao_filter.py:258-266): Savitzky-Golay (order 4) on log(k P) along log k; the last half-window is left untouched."""
    pk = pk.reshape(k.size, -1)
    n = savgol_length(k)
    out = (np.exp(signal.savgol_filter(np.log(k * pk.T), n, polyorder=4, axis=-1)) / k).T
    out[-(n // 2):] = pk[-(n // 2):]
    return out


def ehsavgol(k, pk, pknow_eh):
    """EHNoWiggleSavGol (bao_filter.py:278-286): Savitzky-Golay on the ratio to the EH no-wiggle P(k, z=0) ``pknow_eh`` (nk,)."""
    pk = pk.reshape(k.size, -1)
    return (signal.savgol_filter(pk.T / pknow_eh, savgol_length(k), polyorder=4, axis=-1) * pknow_eh).T


def ehpoly(k, pk, pknow_eh, rs_ratio=1., krange=(1e-3, 1.)):
    """EHNoWigglePoly (bao_filter.py:323-342): the ratio to EH no-wiggle emulated by sum_i a_i k^(i-2), i < 6, constrained at both ends."""
    pk = pk.reshape(k.size, -1)
    kr = np.asarray(krange) / rs_ratio
    mask = (k >= kr[0]) & (k <= kr[1])
    kk = k[mask]
    ratio = pk[mask].T / pknow_eh[mask]
    gradient = np.array([kk**(i - 2) for i in range(6)])
    cg = np.column_stack([gradient[..., 0], gradient[..., 1] - gradient[..., 0], gradient[..., -1], gradient[..., -2] - gradient[..., -1]])
    constraint = np.column_stack([ratio[..., 0], ratio[..., 1] - ratio[..., 0], ratio[..., -1], ratio[..., -2] - ratio[..., -1]])
    model = np.stack([least_squares_constrained(gradient, kk**2, cg, ratio[i], constraint[i]) for i in range(ratio.shape[0])])
    wiggles = np.ones_like(pk)
    wiggles[mask] = (ratio / model).T
    return pk / wiggles


def peakaverage_prepare(k, pk_fid, pknow_fid):
    """PeakAverage._prepare (bao_filter.py:536-563): ``pk_fid`` / ``pknow_fid`` callables of the fiducial cosmology at z = 0."""
    index = np.flatnonzero((k >= 1e-3) & (k <= 1.))
    k_fid = k[index]
    ratio = pk_fid(k_fid) / pknow_fid(k_fid)
    gradient = np.array([k_fid**(i - 1) for i in range(4)])
    cg = np.column_stack([gradient[..., 0], gradient[..., 1] - gradient[..., 0], gradient[..., -1], gradient[..., -2] - gradient[..., -1]])
    corr = least_squares_constrained(gradient, k_fid**2, cg, ratio, [ratio[..., 0], ratio[..., 1] - ratio[..., 0], ratio[..., -1], ratio[..., -2] - ratio[..., -1]])
    ik0 = np.searchsorted(k_fid, 1e-2, side='right') + 1
    k_peaks, pad_peaks = [], []
    for si in [1., -1.]:
        ik = signal.find_peaks(si * ratio[ik0:] / corr[ik0:])[0] + ik0
        npadlow = index[0]
        ik += npadlow
        ikmax = max(index[-1], ik[-1] + 1)
        pad_peaks.append((npadlow, len(ik), k.size - ikmax))
        k_peaks.append(k[np.concatenate([np.arange(npadlow), ik, np.arange(ikmax, k.size)], axis=0)])
    return k_peaks, pad_peaks


def peakaverage(k, pk, pknow_eh, prep, rs_ratio=1.):
    """PeakAverage._interp / _compute (bao_filter.py:565-580): mean of two natural splines (in log10 k) through the ratio to EH no-wiggle
    sampled at the fiducial maxima / minima moved by the rs_drag ratio."""
    from .interp import spline1d
    pk = pk.reshape(k.size, -1)
    k_peaks, pad_peaks = prep
    y = pk / pknow_eh[:, None]
    logx = np.log10(k)
    first = spline1d(logx, y, extrap=True)
    out = 0.
    for kp, npad in zip(k_peaks, pad_peaks):
        rescale = np.concatenate([np.linspace(1., rs_ratio, npad[0]), np.full(npad[1], rs_ratio), np.linspace(rs_ratio, 1., npad[2])])
        logxx = np.log10(kp / rescale)
        out = out + spline1d(logxx, first(logxx))(logx)
    return out / 2. * pknow_eh[:, None]


def _simpson_avg(y, x):
    """The reference's simpson (jax.py:365-507, scipy v1.0.0 rule, even='avg') along the last axis, for an even or odd number of samples."""
    def basic(y, start, stop, x):
        s0, s1, s2 = slice(start, stop, 2), slice(start + 1, stop + 1, 2), slice(start + 2, stop + 2, 2)
        h = np.diff(x)
        h0, h1 = h[s0], h[s1]
        hsum, hprod, h0divh1 = h0 + h1, h0 * h1, h0 / h1
        return np.sum(hsum / 6. * (y[..., s0] * (2. - 1. / h0divh1) + y[..., s1] * (hsum * hsum / hprod) + y[..., s2] * (2. - h0divh1)), axis=-1)
    n = y.shape[-1]
    if n % 2 == 1:
        return basic(y, 0, n - 2, x)
    val = 0.5 * (x[-1] - x[-2]) * (y[..., -1] + y[..., -2]) + basic(y, 0, n - 3, x)
    val = val + 0.5 * (x[1] - x[0]) * (y[..., 1] + y[..., 0]) + basic(y, 1, n - 2, x)
    return val / 2.


def bspline(k, pk, pknow_eh, constraint=('sigma8',)):
    """BSpline (bao_filter.py:622-688): the ratio to the EH no-wiggle P(k, z=0) on 5e-3 <= k <= 1 fitted by B-splines in log10 k of degrees 5, 6
    (, 7) on 14, 14 (, 15) knots, each pinned at its four end samples; the models are then mixed with coefficients that sum to one and
    reproduce sigma8 (, sigma_d) of the input.  The mixing system is solved per column with right-hand sides taken as VECTORS, the meaning
    of ``numpy.linalg.solve(a, b)`` for b.ndim == a.ndim - 1 before numpy 2 -- under numpy 2 the reference's call (:685) raises for any
    constraint, see tests/golden/bspline.npz and oracle/gen_golden.py: gen_bspline."""
    pk = pk.reshape(k.size, -1)
    kmin, kmax = 5e-3, 1.
    mask = (k >= kmin) & (k <= kmax)
    logk = np.log10(k[mask])
    weights = 1 + 1e6 * np.tanh(0.005 * (logk + 1.1)**16)
    weights /= np.sum(weights)
    ratio = pk[mask].T / pknow_eh[mask]
    ends = np.column_stack([ratio[..., 0], ratio[..., 1] - ratio[..., 0], ratio[..., -1], ratio[..., -2] - ratio[..., -1]])
    models = []
    for nknots, degree in [(14, 5), (14, 6), (15, 7)][:1 + len(constraint)]:
        ts = np.concatenate([np.zeros(degree + 1), np.arange(1, nknots - 2 * degree) / (nknots - 2 * degree), np.ones(degree + 1)])
        ts = np.log10((kmax - kmin) * ts + kmin)
        gradient = np.array([interpolate.BSpline(ts, np.eye(len(ts) - degree - 1)[i], degree)(logk) for i in range(nknots - degree)])
        cg = np.column_stack([gradient[..., 0], gradient[..., 1] - gradient[..., 0], gradient[..., -1], gradient[..., -2] - gradient[..., -1]])
        params = np.concatenate([ratio, ends], axis=-1).dot(least_squares_projector(gradient, weights, cg))[..., :gradient.shape[0]]
        model = pk.T.copy()
        model[..., mask] = params.dot(gradient) * pknow_eh[mask]
        models.append(model)
    models = np.array(models)                    # (nmodels, ncol, nk)

    def tophat(x):
        return 3 * (np.sin(x) - x * np.cos(x)) / x**3

    functionals = {'sigma8': lambda p: 1 / (2. * np.pi**2) * _simpson_avg(k**2 * tophat(8. * k)**2 * p, k),
                   'sigmad': lambda p: 1 / (6. * np.pi**2) * _simpson_avg(p, k)}
    out = np.empty_like(pk)
    for ic in range(pk.shape[1]):
        system, target = [np.ones(len(models))], [1.]
        for name in constraint:
            system.append(np.array([functionals[name](m[ic]) for m in models]))
            target.append(functionals[name](pk[:, ic]))
        coeffs = np.linalg.solve(np.array(system), np.array(target))
        out[:, ic] = np.sum(coeffs[:, None] * models[:, ic], axis=0)
    return out
