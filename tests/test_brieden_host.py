"""The arithmetic of brieden_resample_kernel (csrc/cp_bao.hip) restated in numpy, lane by lane, against scipy's natural CubicSpline through the
padded knots (what the reference evaluates: bao_filter.py:500-509 with interpolator.py:42-87): a spline through uniformly spaced knots with two
extrapolated knots on either side, solved by two first-order recursions per segment of S knots, the neighbours' totals over 33 knots, and the
homogeneous corrections A p^i + B p^(n-1-i) in closed form.  CPU only: what the kernel computes is pinned here without a GPU; the kernel itself
is compared with the oracle in tests/test_bao_gpu.py."""
import numpy as np
import pytest
from scipy.interpolate import CubicSpline

P = np.sqrt(3.) - 2.
KAPPA = 1. / (2. * np.sqrt(3.))


def pad_knots(x, y, kmin=1e-7, kmax=1e2):
    """_pad_log on log10 values (interpolator.py:42-87)."""
    lmin = np.log10(min(kmin, 10**x[0] * (1 - 1e-9)))
    lmax = np.log10(max(kmax, 10**x[-1] * (1 + 1e-9)))
    sl, sr = (y[1] - y[0]) / (x[1] - x[0]), (y[-1] - y[-2]) / (x[-1] - x[-2])
    xa, xb, xc, xd = lmin, x[0] * 0.1 + lmin * 0.9, x[-1] * 0.1 + lmax * 0.9, lmax
    return (np.concatenate([[xa, xb], x, [xc, xd]]),
            np.concatenate([[y[0] + sl * (xa - x[0]), y[0] + sl * (xb - x[0])], y, [y[-1] + sr * (xc - x[-1]), y[-1] + sr * (xd - x[-1])]]))


def second_derivatives_as_the_kernel(x, y, X):
    """M at the n + 4 padded knots, in the kernel's steps: S knots per lane, 64 lanes."""
    n = x.size
    S = (n + 63) // 64
    reach = (32 + S) // S
    h = (x[-1] - x[0]) / (n - 1)
    rhs = np.zeros(64 * S)
    rhs[1:n - 1] = (y[2:] - y[1:-1]) - (y[1:-1] - y[:-2])      # the rows of the two end knots: their right-hand sides vanish
    g = rhs.reshape(64, S).copy()                               # lane, knot of the lane
    for t in range(S - 2, -1, -1):
        g[:, t] += P * g[:, t + 1]
    gtot = g[:, 0].copy()                                       # anti-causal total of the segment, seen from its first knot
    f = np.zeros(64)
    for t in range(S):
        e = g[:, t] + P * f
        f = e - P * g[:, t + 1] if t + 1 < S else e
        g[:, t] = e
    fin, gin, w = np.zeros(64), np.zeros(64), 1.
    fl, gr = f.copy(), gtot.copy()
    for _ in range(reach):                                      # the neighbours' totals, lane by lane, weights (p^S)^m: 33 knots
        fl = np.concatenate([[0.], fl[:-1]])
        gr = np.concatenate([gr[1:], [0.]])
        fin += w * fl
        gin += w * gr
        w *= P**S
    for t in range(S):
        g[:, t] += P**(S - t) * gin + P**(t + 1) * fin
    m0 = g.reshape(-1)[:n].copy()

    def amplitude(ha, H, ma, mb):
        alpha = 2. * (H + h) - H * H / (2. * (ha + H))
        return -(alpha * ma + h * mb) / (alpha + h * P)

    A = amplitude(X[1] - X[0], X[2] - X[1], m0[0], m0[1])
    B = amplitude(X[-1] - X[-2], X[-2] - X[-3], m0[-1], m0[-2])
    i = np.arange(n)
    with np.errstate(under='ignore'):
        m = m0 + np.where(i <= 40 + S, A * P**np.minimum(i, 400), 0.) + np.where(n - 1 - i <= 40 + S, B * P**np.minimum(n - 1 - i, 400), 0.)
    full = np.zeros(n + 4)
    full[2:-2] = m
    full[1] = -(X[2] - X[1]) * m[0] / (2. * ((X[1] - X[0]) + (X[2] - X[1])))
    full[-2] = -(X[-2] - X[-3]) * m[-1] / (2. * ((X[-1] - X[-2]) + (X[-2] - X[-3])))
    return full * (6. * KAPPA / h**2)


@pytest.mark.parametrize('n', [341, 129, 200, 512, 450])
@pytest.mark.parametrize('rescale', [0.87, 1., 1.0004, 1.15])
def test_the_scheme_is_the_natural_spline(n, rescale):
    rng = np.random.default_rng(n)
    kf = np.geomspace(1e-3, 1., n)
    x = np.log10(kf / rescale)
    y = np.log10(1e4 * kf / (1. + (kf / 0.02)**2)**1.3 * (1. + 0.03 * np.sin(kf / 0.013 + rng.uniform(0., 6.))))
    X, Y = pad_knots(x, y)
    spline = CubicSpline(X, Y, bc_type='natural')
    m = second_derivatives_as_the_kernel(x, y, X)
    truth = spline(X, 2)
    assert np.abs(m - truth).max() < 2e-12 * np.abs(truth).max()
    xq = np.log10(kf)
    j = np.clip(np.searchsorted(X, xq, side='right') - 1, 0, n + 2)
    hh = X[j + 1] - X[j]
    b = (xq - X[j]) / hh
    a = 1. - b
    v = a * Y[j] + b * Y[j + 1] + ((a**3 - a) * m[j] + (b**3 - b) * m[j + 1]) * hh**2 / 6.
    assert np.abs(v - spline(xq)).max() < 5e-15 * np.abs(Y).max()
