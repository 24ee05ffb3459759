"""Pin the interpolator / P(k) <-> xi(s) oracle (oracle/interp.py) against golden vectors from the reference:
tabulated P(k) (G5 part of tests/golden/sigma.npz) and the xi side (tests/golden/xi.npz)."""
import numpy as np

from oracle import background as ob
from oracle import interp as oi
from oracle import power as op
from oracle import sigma as osg
from test_oracle_power_sigma import eh_default_callable


def tilted(a, b, s, q=1.5):
    """max |a - b| s^q / max |b| s^q; NaNs (queries outside the table) must sit at the same places"""
    w = s**q if np.ndim(a) == 1 else s[:, None]**q
    assert (np.isnan(a) == np.isnan(b)).all()
    return np.nanmax(np.abs((a - b) * w)) / np.nanmax(np.abs(b * w))


def growth_sq(z):
    p = ob.derived()
    return op.growth_factor(np.asarray(z, dtype='f8'), p, znorm=0.)**2


def test_tabulated_pk(golden):
    g = golden('sigma')
    kt, zt, pkt = g['table_k'], g['table_z'], g['table_pk']
    tab = oi.pk_interp_2d(kt, zt, pkt)
    np.testing.assert_allclose(tab(g['table_eval_k'], g['z'][::4]), g['table_eval'], rtol=1e-10, equal_nan=True)
    tab1 = oi.pk_interp_1d(kt, pkt[:, 0])
    np.testing.assert_allclose(tab1(g['table_eval_k']), g['table1d_eval'], rtol=1e-11, equal_nan=True)
    np.testing.assert_allclose(np.sqrt(osg.sigma_r2(g['r'][::8], tab1)), g['table1d_sigma_r'], rtol=1e-10)
    np.testing.assert_allclose(np.sqrt(osg.sigma_r2(g['r'], lambda k: tab(k, g['z']))), g['table_sigma_rz'], rtol=1e-9)


def test_pad_log():
    k = np.logspace(-3, 1, 20)
    pk = 3. * k**-1.2
    lk, lp = oi.pad_log(k, pk)
    assert lk.size == 24 and lk[0] == -7. and lk[-1] == 2.
    np.testing.assert_allclose(lp, np.log10(3.) - 1.2 * lk, rtol=1e-12)       # a power law continues exactly
    lk2, _ = oi.pad_log(k, pk, extrap_kmin=1e-2, extrap_kmax=1.)              # range inside the table: edge * (1 -+ 1e-9)
    np.testing.assert_allclose([10**lk2[0], 10**lk2[-1]], [k[0] * (1 - 1e-9), k[-1] * (1 + 1e-9)], rtol=1e-12)


def test_xi_1d(golden):
    g = golden('xi')
    pk1 = eh_default_callable(0.)
    s, xi = oi.to_xi(pk1)
    np.testing.assert_allclose(s, g['xi1_s'], rtol=1e-13)
    assert tilted(xi, g['xi1_xi'], s) < 1e-13
    xi1 = oi.xi_interp_1d(g['xi1_s'], g['xi1_xi'])
    assert tilted(xi1(g['sq']), g['xi1_eval'], g['sq']) < 1e-12
    oob = xi1(np.array([g['xi1_s'][0] * 0.5, 1., g['xi1_s'][-1] * 2.]))
    assert np.isnan(oob[0]) and np.isnan(oob[2]) and np.isnan(g['xi1_eval_oob'][[0, 2]]).all()
    np.testing.assert_allclose(oob[1], g['xi1_eval_oob'][1], rtol=1e-10)
    k, pk = oi.to_pk(xi1, g['xi1_s'][0], g['xi1_s'][-1])
    np.testing.assert_allclose(k, g['xi1_to_pk_k'], rtol=1e-13)
    assert tilted(pk, g['xi1_to_pk_pk'], k) < 1e-11
    # P(k) back from the default k range rings below zero at the edges: its log is NaN and so is everything derived from it
    assert (pk <= 0).any() and np.isnan(g['xi1_to_pk_eval']).all() and np.isnan(g['xi1_sigma8'])
    assert np.isnan(oi.pk_interp_1d(k, pk)(g['kq'])).all()
    # the reference's own round trip (tests/test_interpolator.py:123-165): k range narrowed to [1e-5, 1e2] first
    c1 = oi.pk_interp_1d(g['c1_k'], g['c1_pk'], extrap_kmin=1e-5, extrap_kmax=1e2)
    s, xi = oi.to_xi(c1, 1e-5, 1e2)
    np.testing.assert_allclose(s, g['xc1_s'], rtol=1e-13)
    assert tilted(xi, g['xc1_xi'], s) < 1e-12
    k, pk = oi.to_pk(oi.xi_interp_1d(s, xi), s[0], s[-1])
    np.testing.assert_allclose(k, g['xc1_to_pk_k'], rtol=1e-13)
    assert tilted(pk, g['xc1_to_pk_pk'], k) < 1e-10
    back = oi.pk_interp_1d(k, pk)
    np.testing.assert_allclose(back(g['kq']), g['xc1_to_pk_eval'], rtol=1e-8)
    np.testing.assert_allclose(np.sqrt(osg.sigma_r2(8., back)), g['xc1_sigma8'], rtol=1e-9)
    np.testing.assert_allclose(np.sqrt(osg.sigma_r2(np.array([2., 8., 30.]), back)), g['xc1_sigma_r'], rtol=1e-9)
    np.testing.assert_allclose(np.sqrt(osg.sigma_d2(back)), g['xc1_sigma_d'], rtol=1e-9)


def test_xi_2d(golden):
    g = golden('xi')
    sq, zq = g['sq'], g['zq']
    z = g['xi2_z']
    s, xi0 = oi.to_xi(eh_default_callable(0.))
    D0sq = growth_sq(0.)
    xi_tab = np.repeat((xi0 / D0sq)[:, None], z.size, axis=1)    # to_xi(ignore_growth=True): the z = 0 shape in every column
    np.testing.assert_allclose(s, g['xi2_s'], rtol=1e-13)
    xi2 = oi.xi_interp_2d(s, z, xi_tab, growth_factor_sq=growth_sq)
    assert tilted(xi2(sq, zq), g['xi2_eval'], sq) < 1e-11
    assert tilted(xi2(sq, zq, ignore_growth=True), g['xi2_eval_nogrowth'], sq) < 1e-11
    np.testing.assert_allclose(xi2(sq[:4], zq, grid=False), g['xi2_eval_pts'], rtol=1e-9)
    # tabulated (s, z) table and 1D tables
    st, zt, tab = g['tab_s'], g['tab_z'], g['tab_xi']
    t2 = oi.xi_interp_2d(st, zt, tab)
    np.testing.assert_allclose(t2(g['tab_sq'], np.array([0.1, 0.9, 1.7])), g['tab2_eval'], rtol=1e-9, atol=1e-14)
    np.testing.assert_allclose(oi.xi_interp_1d(st, tab[:, :2])(g['tab_sq']), g['tab1_eval'], rtol=1e-10, atol=1e-15)
    np.testing.assert_allclose(oi.xi_interp_1d(st, tab[:, 0], interp_s='lin')(g['tab_sq']), g['tab1_lin_eval'], rtol=1e-10, atol=1e-15)
    # 2D round trip on the narrowed range: tabulated (540 x 30) P without growth + growth_factor_sq
    c2 = oi.pk_interp_2d(g['c2_k'], g['c2_z'], g['c2_pk'], extrap_kmin=1e-5, extrap_kmax=1e2, growth_factor_sq=growth_sq)
    np.testing.assert_allclose(c2(g['kq'], zq), g['c2_eval'], rtol=1e-9)
    s, xi = oi.to_xi(lambda k: c2(k, g['c2_z'], ignore_growth=True), 1e-5, 1e2)
    np.testing.assert_allclose(s, g['xc2_s'], rtol=1e-13)
    xc2 = oi.xi_interp_2d(s, g['c2_z'], xi, growth_factor_sq=growth_sq)
    assert tilted(xc2(sq, zq), g['xc2_eval'], sq) < 1e-10
    k, pk = oi.to_pk(lambda ss: xc2(ss, g['c2_z'], ignore_growth=True), s[0], s[-1])
    pc2 = oi.pk_interp_2d(k, g['c2_z'], pk, growth_factor_sq=growth_sq)
    np.testing.assert_allclose(pc2(g['kq'], zq), g['xc2_to_pk_eval'], rtol=1e-8)
    np.testing.assert_allclose(np.sqrt(osg.sigma_r2(8., lambda kk: pc2(kk, zq))), g['xc2_sigma8_z'], rtol=1e-9)
    np.testing.assert_allclose(np.sqrt(osg.sigma_d2(lambda kk: pc2(kk, zq))), g['xc2_sigma_dz'], rtol=1e-9)


def test_kirkby2013(golden):
    """oracle/bao.py: kirkby2013 against the reference filter (default boxes; boxes rescaled by the rs_drag ratio; 2D input)."""
    from oracle import bao as obao
    g = golden('xi')
    s = g['kirkby_s']
    np.testing.assert_allclose(obao.kirkby2013(s, g['kirkby1_xi']), g['kirkby1_xinow'], rtol=1e-9, atol=1e-14)
    np.testing.assert_allclose(obao.kirkby2013(s, g['kirkby1_xi'], rescale=float(g['kirkby1r_ratio'])), g['kirkby1r_xinow'], rtol=1e-9, atol=1e-14)
    np.testing.assert_allclose(obao.kirkby2013(s, g['kirkby1_xi'], rescale=float(g['kirkby1d_ratio'])), g['kirkby1d_xinow'], rtol=1e-9, atol=1e-14)
    assert np.abs(g['kirkby1_xinow'] - g['kirkby1_xi']).max() > 1e-4          # the peak is really removed ...
    m = (s < 80.) | (s > 155.)
    np.testing.assert_array_equal(g['kirkby1_xinow'][m], g['kirkby1_xi'][m])   # ... and nothing else is touched
