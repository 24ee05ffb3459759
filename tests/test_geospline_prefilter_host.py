"""CPU: the mathematics of the prefiltered FFTLog + spline plan (cp_geospline_plan_create_prefiltered, csrc/cp_sigma.hip) without a GPU.  The library's
host-only entry cp_geospline_basis gives the B-spline pieces the kernel evaluates with: held to scipy's BSpline on the same geometric knots; and the whole
route -- the division of the transform's u by conj(alpha / lambda e^{-i w} + beta + gamma lambda e^{i w}), the transform's ordinary arithmetic (the oracle's
numpy restatement of fftlog.py:228-235), four coefficients and four cubic weights per radius -- restated in numpy on those pieces, against scipy's natural
CubicSpline of the ordinary transform's output (what the reference computes: interpolator.py:285-291).  The kernel itself: tests/test_fused_kernels_gpu.py."""
import ctypes

import numpy as np
import pytest
from scipy.interpolate import BSpline, CubicSpline

from cosmoprimo_amd import _lib
from oracle import fftlog as ofl


def basis_of(rho):
    out = np.zeros(16)
    _lib.check(_lib.load().cp_geospline_basis(ctypes.c_double(rho), _lib.as_double_p(out)))
    return out.reshape(4, 4)


def cox_de_boor(knots, r):
    """Value at r of the cubic B-spline on five knots, by the recursion on VALUES in extended precision (the library recurs on polynomial coefficients)."""
    t = np.asarray(knots, dtype=np.longdouble)
    r = np.asarray(r, dtype=np.longdouble)
    b = [np.where((r >= t[i]) & (r < t[i + 1]), np.longdouble(1), np.longdouble(0)) for i in range(4)]
    for d in range(1, 4):
        b = [(r - t[i]) / (t[i + d] - t[i]) * b[i] + (t[i + d + 1] - r) / (t[i + d + 1] - t[i + 1]) * b[i + 1] for i in range(4 - d)]
    return b[0]


@pytest.mark.parametrize('rho', [1.0001, np.exp(np.log(1e9) / 1023), 1.05, 1.7])
def test_pieces_are_the_b_splines_of_the_geometric_knots(rho):
    K = basis_of(rho)
    x = np.linspace(0., 1., 41)[:-1]
    lrho = np.longdouble(rho)
    r = 1 + np.asarray(x, dtype=np.longdouble) * (lrho - 1)
    for i in range(4):      # the B-spline centred on rho^(i - 1): knots rho^(i - 3) ... rho^(i + 1)
        ref = cox_de_boor(lrho**np.arange(i - 3, i + 2), r)
        np.testing.assert_allclose(np.polyval(K[i][::-1], x), np.asarray(ref, dtype='f8'), rtol=0., atol=2e-15)
    # (scipy's own evaluation in double loses 1e-16 / (rho - 1) to the knot differences: it agrees to that)
    ref = BSpline.basis_element(rho**np.arange(-2., 3.), extrapolate=False)(np.asarray(r, dtype='f8'))
    np.testing.assert_allclose(np.polyval(K[1][::-1], x), ref, rtol=0., atol=1e-15 / (rho - 1.) + 2e-15)
    np.testing.assert_allclose(K.sum(axis=0), [1., 0., 0., 0.], atol=1e-13)      # a partition of unity
    assert K[3][0] == 0. and abs(K[:3, 0].sum() - 1.) < 1e-15      # alpha + beta + gamma = 1 at a knot


def test_bad_arguments():
    out = np.zeros(16)
    for rho in (1., 0.5, float('nan'), float('inf')):
        assert _lib.load().cp_geospline_basis(ctypes.c_double(rho), _lib.as_double_p(out)) == _lib.CP_EINVAL
    assert _lib.load().cp_geospline_basis(ctypes.c_double(1.1), None) == _lib.CP_EINVAL


@pytest.mark.parametrize('q', [0., 0.3])
def test_prefiltered_transform_gives_the_natural_spline(q):
    nk = 1024
    k = np.geomspace(1e-7, 1e2, nk)
    pk = 2e4 * (k / 0.02)**0.96 / (1 + (k / 0.02)**2)**1.7 * (1 + 0.05 * np.sin(k / 0.01) * np.exp(-(k / 0.3)**2))
    plain = ofl.tophat_variance(k, q=q)
    var = ofl.apply(plain, pk[None, None, :])[0, 0]
    s = plain.y[0]
    npad, rho = plain.npad, float(np.exp(plain.delta[0]))
    post = plain.post[0]
    lam = (post[-1] / post[0])**(1. / (npad - 1))
    np.testing.assert_allclose(post[1:] / post[:-1], lam, rtol=1e-11)      # the power law the plan insists on
    K = basis_of(rho)
    alpha, beta, gamma = K[0][0] / lam, K[1][0], K[2][0] * lam
    w = 2. * np.pi * np.arange(npad // 2 + 1) / npad
    filtered = ofl.tophat_variance(k, q=q)
    filtered.u = plain.u / np.conj(alpha * np.exp(-1j * w) + beta + gamma * np.exp(1j * w))
    c = ofl.apply(filtered, pk[None, None, :], keep_padding=True)[0, 0]      # the coefficient sequence on the padded grid
    full = ofl.apply(plain, pk[None, None, :], keep_padding=True)[0, 0]
    # the interpolation conditions hold on the whole periodic grid, to the rounding of the transform (tilted space: the FFT's own scale)
    lhs = K[0][0] * np.roll(c, 1) + K[1][0] * c + K[2][0] * np.roll(c, -1)
    tilt = 1. / np.abs(post)
    assert np.abs((lhs - full) * tilt)[1:-1].max() < 1e-13 * np.abs(full * tilt).max()
    # evaluation as the kernel does it, at radii at least 32 knots inside the grid (the plan refuses the others)
    r = np.geomspace(s[32] * 1.0001, s[-33] * 0.9999, 1000)
    j = np.searchsorted(s, r, side='right') - 1
    x = (r - s[j]) / (s[j + 1] - s[j])
    off = plain.out_left
    val = sum(c[off + j - 1 + i] * np.polyval(K[i][::-1], x) for i in range(4))
    ref = CubicSpline(s, var, bc_type='natural')(r)
    tilted = np.abs(val - ref) * r**(1.5 + q)
    assert tilted.max() < 1e-13 * np.abs(var * s**(1.5 + q)).max()
    # ... and what the periodic ends cost nearer to the ends of the knots: forgotten like 0.27^distance
    near = np.array([s[8] * 1.01])
    jn = np.searchsorted(s, near, side='right') - 1
    xn = (near - s[jn]) / (s[jn + 1] - s[jn])
    vn = sum(c[off + jn - 1 + i] * np.polyval(K[i][::-1], xn) for i in range(4))
    rn = CubicSpline(s, var, bc_type='natural')(near)
    assert 1e-13 < abs(vn / rn - 1.)[0] < 1e-3      # (why radii within 32 knots of the ends take the other kernel)
