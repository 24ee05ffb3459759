"""Pin the numpy oracle (oracle/fftlog.py) against golden vectors produced by the reference (G1, G3)."""
import numpy as np
import pytest

from oracle import fftlog as ofl
from conftest import tilted_err

CASES = {
    'p2c_l0': lambda k: ofl.power_to_correlation(k, ell=0),
    'p2c_multi': lambda k: ofl.power_to_correlation(k, ell=[0, 1, 2, 3, 4]),
    'c2p_l0': lambda k: ofl.correlation_to_power(k, ell=0),
    'c2p_l2_q': lambda k: ofl.correlation_to_power(k, ell=2, q=0.5),
    'tophat': lambda k: ofl.tophat_variance(k),
    'gauss': lambda k: ofl.gaussian_variance(k),
    'hankel_nu0_q1': lambda k: ofl.hankel(k, nu=0, q=1),
    'hankel_nu2': lambda k: ofl.hankel(k, nu=[0, 2], q=1),
    'p2c_nolowring': lambda k: ofl.power_to_correlation(k, ell=0, lowring=False, xy=1.),
}


def check_tables(t, g, prefix, every):
    ev4 = max(every // 4, 1)
    assert np.array_equal(g[prefix + 'sizes'], [t.npad, t.in_left, t.in_right, t.out_left, t.out_right])
    np.testing.assert_allclose(t.delta, g[prefix + 'delta'], rtol=1e-15)
    np.testing.assert_allclose(t.lnxy, g[prefix + 'lnxy'], rtol=1e-13, atol=1e-18)
    np.testing.assert_allclose(t.y[..., ::ev4], g[prefix + 'y'], rtol=1e-14)
    np.testing.assert_allclose(t.u[..., ::ev4], g[prefix + 'u'], rtol=1e-12, atol=1e-300)
    np.testing.assert_allclose(t.pre[..., ::every], g[prefix + 'pre'], rtol=1e-13)
    np.testing.assert_allclose(np.real(t.post[..., ::every]), np.real(g[prefix + 'post']), rtol=1e-13)


@pytest.mark.parametrize('n', [1024, 2048])
@pytest.mark.parametrize('name', sorted(CASES))
def test_tables(golden, n, name):
    g = golden('fftlog_tables')
    k = np.logspace(-5, 2, n)
    t = CASES[name](k)
    check_tables(t, g, 'n%d_%s_' % (n, name), 1 if name in ('p2c_l0', 'tophat') else 16)


def test_pad7(golden):
    # reference tests/test_fftlog.py:40-53 (minfolds=3, lowring=False, xy=1)
    g = golden('fftlog_tables')
    x = np.logspace(-3, 3, num=7, endpoint=True)
    t = ofl.hankel(x, nu=0, minfolds=3, lowring=False, xy=1.)
    check_tables(t, g, 'pad7_', 1)
    np.testing.assert_allclose(t.padded_x, np.logspace(-15, 16, num=32)[None, :])
    np.testing.assert_allclose(t.padded_y, np.logspace(-16, 15, num=32)[None, :])
    np.testing.assert_allclose(t.padded_x, g['pad7_padded_x'], rtol=1e-14)


def test_pad_identities():
    # reference tests/test_fftlog.py:28-38
    a = np.ones((6, 6))
    pa = np.zeros((6, 13)); pa[:, 3:9] = 1
    assert np.allclose(ofl.pad(a, (3, 4), 0), pa)
    assert np.allclose(ofl.pad(a, (4, 3), 'edge'), np.ones((6, 13)))
    c = np.array([(i + 1) * np.logspace(-3, 3, num=6, endpoint=False) for i in range(3)])
    pc = np.array([(i + 1) * np.logspace(-12, 12, num=24, endpoint=False) for i in range(3)])
    assert np.allclose(ofl.pad(c, (9, 9), 'log'), pc)


@pytest.mark.parametrize('n', [1024, 2048])
def test_transforms(golden, n):
    g = golden('fftlog_transforms')
    pkd = golden('pk_eh_default')
    k, pk = pkd['k%d' % n], pkd['pk%d' % n]
    t = ofl.power_to_correlation(k, ell=0)
    s = g['n%d_p2c_l0_s' % n]
    np.testing.assert_allclose(t.y[0], s, rtol=1e-14)
    for name, extrap in [('zero', 0), ('edge', 'edge'), ('log', 'log'), ('mixed', ('log', 0.)), ('const', (1.5, 'edge'))]:
        xi = ofl.apply(t, pk, extrap=extrap)[0]
        assert tilted_err(xi, g['n%d_p2c_l0_%s' % (n, name)], s, 1.5) < 1e-14
    keep = ofl.apply(t, pk, extrap='log', keep_padding=True)[0]
    assert tilted_err(keep, g['n%d_p2c_l0_keep' % n], t.padded_y[0], 1.5) < 1e-14
    tm = ofl.power_to_correlation(k, ell=[0, 2, 4])
    xi = ofl.apply(tm, pk)
    assert xi.shape == (3, n)
    for i in range(3):
        assert tilted_err(xi[i], g['n%d_p2c_l024' % n][i], tm.y[i], 1.5) < 1e-14
    tt = ofl.tophat_variance(k)
    assert tilted_err(ofl.apply(tt, pk)[0], g['n%d_tophat' % n], tt.y[0], 1.5) < 1e-14
    tc = ofl.correlation_to_power(s, ell=0)
    assert tilted_err(ofl.apply(tc, g['n%d_p2c_l0_zero' % n])[0], g['n%d_c2p_l0' % n], tc.y[0], 1.5) < 1e-13


def test_config2_rows(golden):
    from oracle.workloads import config2_rows
    g = golden('fftlog_transforms')
    pkd = golden('pk_eh_default')
    k, pk = pkd['k2048'], pkd['pk2048']
    t = ofl.power_to_correlation(k, ell=0)
    rows = np.concatenate([config2_rows(k, pk, i, i + 1) for i in g['config2_idx']])
    xi = ofl.apply(t, rows[:, None, :])[:, 0]
    for a, b in zip(xi, g['config2_xi']):
        assert tilted_err(a, b, t.y[0], 1.5) < 1e-14


def test_hankel_pair(golden):
    # analytic pair f=(1+x^2)^-1.5 <-> g=exp(-y), reference tests/test_fftlog.py:58-81
    g = golden('fftlog_transforms')
    x = g['hankel60_x']
    f = 1 / (1 + x**2)**1.5
    t = ofl.hankel(x, nu=0, q=1)
    out = ofl.apply(t, f, extrap='log')[0]
    assert np.allclose(out, np.exp(-t.y[0]), rtol=1e-8, atol=1e-8)
    np.testing.assert_allclose(out, g['hankel60_g'], rtol=1e-11, atol=1e-14)
    ti = ofl.inverse(t)
    check_tables(ti, g, 'hankel60_inv_', 1)
    f2 = ofl.apply(ti, out, extrap='log')[0]
    assert np.allclose(f2, f, rtol=1e-7, atol=1e-7)
    np.testing.assert_allclose(f2, g['hankel60_inv_f'], rtol=1e-10, atol=1e-13)
    # reverse direction, batched (3, 60) input
    y = np.logspace(-4, 2, num=60, endpoint=False)
    tg = ofl.hankel(y, nu=0, q=1)
    scales = np.linspace(1., 3., 3)
    fb = ofl.apply(tg, (np.exp(-y) * scales[:, None])[:, None, :], extrap='log')[:, 0]
    assert np.allclose(fb / scales[:, None], 1 / (1 + tg.y[0]**2)**1.5, rtol=1e-10, atol=1e-10)


def test_generic_kernels(golden):
    g = golden('fftlog_tables')
    x = np.logspace(-4, 3, 200)
    kern = {
        'tophat1': lambda z: ofl.u_tophat(z, 1), 'tophat3': lambda z: ofl.u_tophat(z, 3), 'tophatsq1': lambda z: ofl.u_tophat_sq(z, 1),
        'tophatsq2': lambda z: ofl.u_tophat_sq(z, 2), 'gaussian': ofl.u_gaussian, 'besselj1': lambda z: ofl.u_bessel_j(z, 1.5),
        'sphbesselj3': lambda z: ofl.u_spherical_bessel_j(z, 3),
    }
    for name, kf in kern.items():
        t = ofl.setup(x[None, :], [kf], [0.7], minfolds=3)
        check_tables(t, g, 'gen200_%s_' % name, 1)
