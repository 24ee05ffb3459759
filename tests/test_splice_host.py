"""CPU: the plan builder of the uniform-stretch spliced spline (csrc/cp_splice_uniform_plan.h, compiled with g++: tests/host_emu/emu_splice.cpp) --
its windows, carry factors and weights against a dense solve of scipy's clamped CubicSpline system, and the kernel's arithmetic (two first-order
recursions per lane over the uniform stretch, junction sums over differences, evaluation) restated in numpy on those tables against scipy itself
(reference bao_filter.py:415-431).  The kernel proper is tested on the GPU (tests/test_fused_kernels_gpu.py::test_spliced_clamped_spline)."""
import ctypes
import os
import subprocess

import numpy as np
import pytest
from scipy.interpolate import CubicSpline

from conftest import ROOT

P = np.sqrt(3.) - 2.


def _lib():
    src = os.path.join(ROOT, 'tests', 'host_emu', 'emu_splice.cpp')
    out = os.path.join(ROOT, 'tests', 'host_emu', 'libemu_splice.so')
    dep = os.path.join(ROOT, 'cosmoprimo_amd', 'csrc', 'cp_splice_uniform_plan.h')
    if not os.path.isfile(out) or os.path.getmtime(out) < max(os.path.getmtime(src), os.path.getmtime(dep)):
        subprocess.check_call(['g++', '-O1', '-std=c++17', '-shared', '-fPIC', '-o', out, src])
    return ctypes.CDLL(out)


def build(knots, pieces, xq):
    """The tables for these knots (pieces: (source, start column, count)) and queries, or None where the scheme does not fit."""
    n, nq = knots.size, xq.size
    first = np.cumsum([0] + [p[2] for p in pieces[:-1]]).astype(np.int32)
    src, start = (np.array([p[i] for p in pieces], dtype=np.int32) for i in (0, 1))
    qj = np.where((xq >= knots[0]) & (xq <= knots[-1]), np.clip(np.searchsorted(knots, xq, side='right') - 1, 0, n - 2), -1).astype(np.int32)
    h = np.diff(knots)
    # the longest run of equal spacings, and the queries that do not simply return their own column of array 0 (what cp_splice_plan_create hands over)
    best = (0, 0)
    lo = 0
    while lo < n - 1:
        hi = lo + 1
        while hi < n - 1 and abs(h[hi] - h[lo]) <= 2e-11 * h[lo]:
            hi += 1
        if hi - lo > best[1] - best[0]:
            best = (lo, hi)
        lo = hi
    col = np.concatenate([np.where(np.full(p[2], p[0] == 0), p[1] + np.arange(p[2]), -1) for p in pieces])      # column of array 0 a knot comes from
    generic = [q for q in range(nq) if not (qj[q] >= 0 and ((xq[q] == knots[qj[q]] and col[qj[q]] == q) or (xq[q] == knots[qj[q] + 1] and col[qj[q] + 1] == q)))]
    gf, ge = (min(generic), max(generic) + 1) if generic else (0, 0)
    ints, doubles, win = np.zeros(15, dtype=np.int32), np.zeros(2), np.zeros((8, 64))
    qe, qw = np.zeros(512, dtype=np.int32), np.zeros((512, 4))
    ip = lambda a: a.ctypes.data_as(ctypes.POINTER(ctypes.c_int))      # noqa: E731
    dp = lambda a: a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))   # noqa: E731
    ok = _lib().emu_splice_build(n, dp(knots), len(pieces), ip(first), ip(src), ip(start), nq, dp(xq), ip(qj), best[0], best[1], gf, ge, ip(ints), dp(doubles),
                                 dp(win), ip(qe), dp(qw))
    if not ok:
        return None
    names = ('S', 'nm', 'wl', 'wr', 'src_u', 'col_u', 'src_l', 'col_l', 'src_r', 'col_r', 'gb0', 'ngb', 'gfirst', 'gend', 'lane_b')
    T = dict(zip(names, (int(v) for v in ints)))
    T.update(mb0=doubles[0], mb1=doubles[1], win=win, qe=qe[:64 * T['ngb']], qw=qw[:64 * T['ngb']], u0=best[0], u1=best[1])
    return T


def run(T, rows):
    """cp_splice_uniform.h restated: rows = (array 0, array 1) of ONE vector -> the spline at the queries (the others return their own column)."""
    S, nm, wl, wr = T['S'], T['nm'], T['wl'], T['wr']
    yu = rows[T['src_u']][T['col_u']:T['col_u'] + nm]
    gl = rows[T['src_l']][T['col_l']:T['col_l'] + wl] if wl else np.zeros(0)
    gr = rows[T['src_r']][T['col_r']:T['col_r'] + wr] if wr else np.zeros(0)
    # second differences of the stretch continued by its end values; lanes x knots
    ext = np.concatenate([yu[:1], yu, np.full(64 * S - nm + 1, yu[-1])])
    r = (ext[2:] - ext[1:-1]) - (ext[1:-1] - ext[:-2])
    r = r.reshape(64, S)
    g = np.zeros((64, S + 1))
    for t in range(S - 1, -1, -1):
        g[:, t] = r[:, t] + P * g[:, t + 1]
    e = np.zeros((64, S))
    f = np.zeros(64)
    for t in range(S):
        e[:, t] = g[:, t] + P * f
        f = e[:, t] - P * g[:, t + 1]
    # the four sums over the differences to the junction knots
    lane = np.arange(64)
    vl = np.concatenate([gl, np.zeros(64 - wl)])
    vr = np.concatenate([gr, np.zeros(64 - wr)])
    ul = np.where(lane < 40, yu[np.minimum(lane, 39)], yu[0])
    ur = np.where(lane < 40, yu[nm - 1 - np.minimum(lane, 39)], yu[0])
    w = T['win']
    gl_last, gr_first = (gl[-1] if wl else 0.), (gr[0] if wr else 0.)
    sums = [np.sum(w[0] * (vl - yu[0]) + w[1] * (ul - yu[0])), np.sum(w[2] * (vl - gl_last) + w[3] * (ul - gl_last)),
            np.sum(w[4] * (ur - yu[-1]) + w[5] * (vr - yu[-1])), np.sum(w[6] * (ur - gr_first) + w[7] * (vr - gr_first))]
    gin = np.concatenate([g[1:, 0], [0.]])
    gin[T['lane_b']] += T['mb0'] * sums[2]
    gin[T['lane_b'] - 1] += T['mb1'] * sums[2]
    fin = np.concatenate([[0.], f[:-1]])
    fin[0] += sums[0]
    t = np.arange(S)
    m = (e + P**(S - t)[None, :] * gin[:, None] + P**(t + 1)[None, :] * fin[:, None]).ravel()
    buf = np.concatenate([[sums[1]], m[:nm], [sums[3]]])      # second derivatives (unscaled) at knots -1 .. nm of the stretch
    yy = np.concatenate([[gl_last], yu, [gr_first]])
    out = rows[0].copy()
    for i in range(64 * T['ngb']):
        q = 64 * T['gb0'] + i
        if T['gfirst'] <= q < T['gend']:
            ee, ww = T['qe'][i], T['qw'][i]
            out[q] = ww[0] * yy[ee + 1] + ww[1] * yy[ee + 2] + (ww[2] * buf[ee + 1] + ww[3] * buf[ee + 2])
    return out


def grids(nk=1024, kmin=1e-7, kmax=1e2, nlin=4096, lin_max=2., lo=1e-2, hi=1.5, left=5e-4, right=2.):
    k = np.geomspace(kmin, kmax, nk)
    klin = np.linspace(kmin, lin_max, nlin)
    mask, ml, mr = (klin > lo) & (klin < hi), k < left, k > right
    knots = np.concatenate([k[ml], klin[mask], k[mr]])
    pieces = [(0, 0, int(ml.sum())), (1, int(np.flatnonzero(mask)[0]), int(mask.sum())), (0, int(np.flatnonzero(mr)[0]), int(mr.sum()))]
    return k, klin, mask, ml, mr, knots, pieces


@pytest.mark.parametrize('kw', [{}, dict(nk=512, kmin=1e-6, kmax=50., nlin=3600, lo=2e-2, hi=1.8, left=1e-3), dict(nk=640, kmin=1e-6, kmax=50., nlin=3500, lin_max=3., lo=2e-2, hi=2.7, left=1e-3, right=3.)])
def test_tables_and_arithmetic_of_the_uniform_stretch_scheme(kw):
    k, klin, mask, ml, mr, knots, pieces = grids(**kw)
    T = build(knots, pieces, k)
    assert T is not None
    nm = int(mask.sum())
    assert T['nm'] == nm and T['S'] % 2 == 1 and 64 * T['S'] >= nm and 64 * (T['S'] - 4) < nm
    assert T['wl'] == min(44, int(ml.sum())) and T['wr'] == min(44, int(mr.sum())) and T['src_u'] == 1 and T['col_u'] == int(np.flatnonzero(mask)[0])
    assert T['col_l'] == int(ml.sum()) - T['wl'] and T['col_r'] == int(np.flatnonzero(mr)[0])
    tb = (nm - 1) - T['S'] * T['lane_b']
    assert 0 <= tb < T['S'] and np.isclose(T['mb0'], P**(tb - T['S']), rtol=1e-12) and np.isclose(T['mb1'], P**tb, rtol=1e-12)
    # each set of weights sums to zero (a constant has no second derivative): what lets the kernel sum over differences
    for a, b in ((0, 1), (2, 3), (4, 5), (6, 7)):
        assert abs(T['win'][a].sum() + T['win'][b].sum()) < 1e-12 * np.abs(T['win'][[a, b]]).max()
    rng = np.random.default_rng(3)
    shape = lambda x: x / (1. + (x / 0.02)**2.6)      # noqa: E731
    for noise in (1e-3, 0.):
        amp = rng.uniform(0.5, 2.)
        pk = amp * shape(k) * (1. + 0.05 * np.sin(k / 0.01) * np.exp(-(k / 0.3)**2))
        lin = amp * shape(klin) * (1. + noise * rng.normal(size=klin.size))
        ref = CubicSpline(knots, np.concatenate([pk[ml], lin[mask], pk[mr]]), bc_type='clamped')(k)
        got = run(T, (pk, lin))
        # (smooth spectra: the uniform grid stands for a linspace whose spacings differ by rounding, 1e-10 on second derivatives eight orders below the largest)
        np.testing.assert_allclose(got, ref, rtol=1e-11 if noise else 1e-9)


def test_plans_the_scheme_does_not_fit():
    # spline queries far from the uniform stretch
    xk = np.concatenate([np.geomspace(1e-4, 9e-3, 200), np.linspace(1e-2, 1., 1500)])
    assert build(xk, [(0, 0, xk.size)], np.geomspace(2e-4, 0.9, 300)) is None
    # no long uniform stretch
    x = np.sort(np.random.default_rng(1).uniform(0., 10., 700))
    assert build(x, [(0, 0, 700)], np.linspace(0.5, 9.5, 200)) is None
    # too many knots on the stretch for 57 per lane
    k, klin, mask, ml, mr, knots, pieces = grids(nlin=8192, hi=1.9)
    assert build(knots, pieces, k) is None
