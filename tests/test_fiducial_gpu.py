"""Fiducial cosmologies (reference cosmoprimo/fiducial.py) end to end: the DESI fiducial (one 0.06 eV-like massive species) through the
massive-neutrino background kernels against the reference's own tabulated DESI cosmology (cosmoprimo/data/desi.dat, computed with a
Boltzmann code; 161 of its 40 002 rows kept in tests/golden/desi_table.npz) and against values produced by the reference in this container."""
import warnings

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_desi_fiducial(golden):
    import cosmoprimo_amd as cp
    from cosmoprimo_amd.fiducial import DESI, BOSS, Planck2018FullFlatLCDM, Uchuu, AbacusSummit
    warnings.simplefilter('ignore')
    g = golden('desi_table')
    cosmo = DESI()
    assert cosmo['N_ncdm'] == 1 and abs(cosmo['omega_ncdm_tot'] - 0.00064420) < 1e-12 and abs(cosmo['N_ur'] - 2.0328) < 1e-12
    ba = cosmo.get_background()
    z = g['z']
    # E(z): same physics as the Boltzmann code to 1e-6 (the radiation / neutrino content matters at high z)
    np.testing.assert_allclose(ba.efunc(z), g['efunc'], rtol=2e-6)
    # D_C: the reference's own statement for this quantity is 1e-6 for z > 0.1 and 4e-4 ... 2e-3 below (natural-spline end condition of its
    # 119-knot table); its interpolation range ends at z = 9999
    dc = ba.comoving_radial_distance(z)
    hi = (z > 0.1) & (z < 9000.)
    np.testing.assert_allclose(dc[hi], g['comoving_radial_distance'][hi], rtol=2e-6)
    lo = (z > 1e-4) & (z <= 0.1)
    np.testing.assert_allclose(dc[lo], g['comoving_radial_distance'][lo], rtol=3e-3)
    # sound horizon of the BBKS / analytic background at a drag redshift (reference tests/test_cosmology.py::test_rs): finite, ~ 147 Mpc
    rs = DESI(engine='bbks').get_background().rs(1059.94)
    assert 140. < rs / cosmo['h'] < 155.
    assert 0.0103 < cosmo['theta_cosmomc'] < 0.0105
    # P(k): sigma8 of the A_s-normalised EH97 spectrum is close to the table's 0.808
    assert abs(cosmo.get_fourier().sigma8_m / 0.807952 - 1.) < 0.05
    for factory in (BOSS, Planck2018FullFlatLCDM):
        c = factory()
        assert c['N_ncdm'] == 1 and abs(c.get_fourier().sigma8_m / c['sigma8'] - 1.) < 1e-6
    assert Uchuu('Planck2018DDE')['w0_fld'] == -0.45 and abs(Uchuu()['Omega_m'] - 0.3089) < 1e-12
    assert abs(DESI(h=0.7)['h'] - 0.7) < 1e-15
    from cosmoprimo_amd.fiducial import DESIDR2Flatw0waCDM
    dr2 = DESIDR2Flatw0waCDM()
    assert abs(dr2['Omega_m'] - 0.3191980194) < 1e-12 and dr2['w0_fld'] == -0.7536302620 and abs(dr2['h'] - 0.6673428704) < 1e-15 and dr2['N_ncdm'] == 1
    assert np.isfinite(dr2.comoving_radial_distance(1.))


def test_tabulated(golden, tmp_path):
    """The 'tabulated' engine (reference tabulated.py, tests/test_tabulated.py): bit-identical to numpy.interp on a table file; TabulatedDESI
    against the reference's own table (161 of its rows, tests/golden/desi_table.npz)."""
    import torch
    from cosmoprimo_amd import Cosmology, CosmologyError
    from cosmoprimo_amd.fiducial import DESI, TabulatedDESI, save_TabulatedDESI
    warnings.simplefilter('ignore')
    g = golden('desi_table')
    fn = str(tmp_path / 'table.dat')
    np.savetxt(fn, np.array([g['z'], g['efunc'], g['comoving_radial_distance']]).T, fmt='%.18e', header='z efunc comoving_radial_distance', comments='# ')
    cosmo = DESI(engine='tabulated', extra_params={'filename': fn, 'names': ['efunc', 'comoving_radial_distance']})
    table = np.loadtxt(fn, unpack=True)
    rng = np.random.default_rng(3)
    z = np.concatenate([10.**rng.uniform(-8, 2, 100000), table[0], [0., 100.], np.nextafter(table[0][1:], 0.), np.nextafter(table[0][:-1], 1e3)])
    for name, column in [('efunc', table[1]), ('comoving_radial_distance', table[2])]:
        out = getattr(cosmo.get_background(), name)(z)
        assert np.array_equal(out, np.interp(z, table[0], column)), name          # same arithmetic as numpy.interp, bit for bit
        assert getattr(cosmo, name)(0.5).shape == () and getattr(cosmo, name)(np.zeros((2, 3), dtype='f4')).dtype == np.float32
    zt = torch.as_tensor(z, device='cuda:0')
    out = cosmo.comoving_radial_distance(zt)
    assert out.is_cuda and np.array_equal(out.cpu().numpy(), np.interp(z, table[0], table[2]))
    z32 = z.astype('f4')
    z32 = z32[(z32 >= table[0][0]) & (z32 <= table[0][-1])]
    out = cosmo.comoving_radial_distance(torch.as_tensor(z32, device='cuda:0'))      # a float32 catalogue on the device stays float32
    assert out.is_cuda and out.dtype == torch.float32 and np.array_equal(out.cpu().numpy(), np.interp(z32.astype('f8'), table[0], table[2]).astype('f4'))
    for bad in (-1., 100.1, [0.5, 200.]):
        with pytest.raises(CosmologyError):
            cosmo.comoving_radial_distance(bad)
    assert cosmo.comoving_radial_distance([]).shape == (0,)

    tab = TabulatedDESI()                                             # reference test_tabulated.py::test_desi
    fn = str(tmp_path / 'cosmo.json')
    tab.write(fn)
    back = Cosmology.read(fn)
    assert np.allclose(back['omega_ncdm'], 0.0006442) and back['N_ncdm'] == 1 and back.engine.name == 'tabulated'
    with pytest.raises(CosmologyError):
        tab.comoving_radial_distance(-1)
    rows = g['z'] <= 100.
    assert np.allclose(tab.efunc(g['z'][rows]), g['efunc'][rows], rtol=2e-6, atol=0.)
    # the reference's file (a Boltzmann code's own background table, interpolated) is off by up to 3e-5 below z = 1e-2: at z = 1e-8 it holds
    # 2.99800696e-05 Mpc/h where c z / (100 km/s/Mpc) = 2.99792458e-05
    dc, ref, zz = tab.comoving_radial_distance(g['z'][rows]), g['comoving_radial_distance'][rows], g['z'][rows]
    assert np.allclose(dc[zz > 1e-2], ref[zz > 1e-2], rtol=3e-6, atol=0.) and np.allclose(dc, ref, rtol=5e-5, atol=1e-10)
    assert abs(dc[1] / (299792.458 / 100. * 1e-8) - 1.) < 1e-7
    z = np.linspace(0, 10, 100)
    assert np.allclose(tab.comoving_radial_distance(z), DESI().comoving_radial_distance(z), rtol=2e-3, atol=1e-10)     # (the 119-knot spline below z = 0.1)
    assert np.allclose(tab.comoving_radial_distance(z[2:]), DESI().comoving_radial_distance(z[2:]), rtol=3e-6)      # (the spline's own 1e-6)
    z0, dz = np.linspace(0, 9.9, 100), 1e-6
    dcdz = (tab.comoving_radial_distance(z0 + dz) - tab.comoving_radial_distance(z0)) / dz
    assert np.allclose(dcdz, 299792.458 / (100. * tab.efunc(z0)), rtol=1e-2, atol=1e-10)
    fn = str(tmp_path / 'desi.dat')
    save_TabulatedDESI(fn)
    again = DESI(engine='tabulated', extra_params={'filename': fn})
    assert np.array_equal(again.comoving_radial_distance(z), tab.comoving_radial_distance(z)) and open(fn).readline().startswith('# z = [0] + np.logspace(-8, 2, 40001)')


def test_abacus(golden):
    """reference tests/test_fiducial.py::test_abacus, test_planck, test_boss, test_uchuu with this package's default engine, and the derived
    parameters the reference itself gets for four of the cosmologies (tests/golden/abacus.npz)."""
    from cosmoprimo_amd import fiducial
    from cosmoprimo_amd.fiducial import AbacusSummit_params, AbacusSummit
    warnings.simplefilter('ignore')
    assert fiducial.Planck2018FullFlatLCDM()['h'] == 0.6766 and fiducial.BOSS()['h'] == 0.676
    cosmo = fiducial.Uchuu()
    for name, value in {'Omega_m': 0.3089, 'h': 0.6774, 'sigma8': 0.8159, 'Omega_b': 0.0486, 'n_s': 0.9667}.items():
        assert abs(cosmo[name] - value) < 1e-15, name
    dcosmos = AbacusSummit_params(params=['root', 'omega_b', 'omega_cdm', 'h', 'A_s', 'n_s', 'alpha_s', 'N_ur', 'omega_ncdm', 'w0_fld', 'wa_fld'])
    assert len(dcosmos) == 98
    assert AbacusSummit_params(19)['omega_ncdm'] == (0.0006442, 0.0006442)
    assert list(AbacusSummit_params(19, params=['h']).keys()) == ['h']
    assert list(AbacusSummit_params(19, params=['omega_k', 'h']).keys()) == ['omega_k', 'h']
    base = AbacusSummit()
    for dcosmo in dcosmos:
        cosmo = AbacusSummit(dcosmo['root'])
        dcosmo.pop('root')
        cosmo2 = base.clone(T_ncdm_over_cmb=None, **dcosmo)
        assert np.allclose(cosmo2._params['N_ur'], cosmo._params['N_ur'])
        assert cosmo == cosmo2.clone(N_eff=cosmo2['N_eff'])
    with pytest.raises(ValueError):
        AbacusSummit('0')
    g = golden('abacus')
    for name in ['000', '009', '019', '130']:
        cosmo = AbacusSummit(name)
        mine = [cosmo['h'], cosmo['Omega_m'], cosmo['N_ur'], cosmo['N_eff'], cosmo['m_ncdm_tot'], cosmo['N_ncdm'], cosmo['w0_fld'], cosmo['wa_fld'],
                cosmo.get_primordial().A_s, cosmo.comoving_radial_distance(1.)]
        np.testing.assert_allclose(np.array(mine, dtype='f8'), g['c' + name], rtol=1e-9, atol=1e-12, err_msg=name)


def test_interp_table_laws():
    """``cp_interp_table``: the interval of a sample guessed from the law of the knots (uniform in x; uniform in log x behind leading knots, the shape of
    data/desi.dat; neither: bisection) and walked to numpy's interval -- bit-identical to numpy.interp in every case, repeated knots and knot hits included."""
    import torch
    from cosmoprimo_amd.tabulated import _InterpTable
    from cosmoprimo_amd import _device as dv
    dev = torch.device('cuda:0')
    rng = np.random.default_rng(11)
    tables = {'uniform': (np.linspace(-2., 5., 3001), 1, 0),
              'desi': (np.concatenate([[0.], np.logspace(-8, 2, 40001)]), 2, 1),
              'geometric': (np.geomspace(3e-4, 7e5, 777), 2, 0),
              'three leading knots': (np.concatenate([[-1., 0., 1e-12], np.geomspace(1e-6, 10., 500)]), 2, 3),
              'irregular': (np.sort(rng.uniform(0., 10., 5000)), 0, 0),
              'nearly uniform (off by a third of a step)': (np.linspace(0., 1., 200) + np.r_[0., 0.33 / 199 * np.sin(np.arange(1, 199)), 0.], 0, 0),
              'repeated knots': (np.repeat(np.linspace(0., 1., 50), 2), 0, 0),
              'two rows': (np.array([1., 2.]), 0, 0)}
    for name, (x, law, first) in tables.items():
        f = np.cos(3. * np.arange(x.size)) * np.sqrt(1. + np.arange(x.size))
        table = _InterpTable(x, f, dev)
        assert table.law == (law, first), (name, table.law)
        inside = rng.uniform(x[0], x[-1], 50000)
        if x[0] > 0:
            inside = np.concatenate([inside, np.exp(rng.uniform(np.log(x[0]), np.log(x[-1]), 50000))])
        q = np.concatenate([inside, x, np.nextafter(x[1:], -np.inf), np.nextafter(x[:-1], np.inf), 0.5 * (x[1:] + x[:-1])])
        q = np.clip(q, x[0], x[-1])
        tq = torch.as_tensor(q, device=dev)
        out = torch.empty_like(tq)
        assert table(tq, out, dv.stream_of(dev)) is False, name
        assert np.array_equal(out.cpu().numpy(), np.interp(q, x, f)), name
        if law:      # float32 samples on the device: computed in double, rounded once
            q32 = q.astype('f4')
            q32 = q32[(q32 >= x[0]) & (q32 <= x[-1])]
            t32 = torch.as_tensor(q32, device=dev)
            out32 = torch.empty_like(t32)
            assert table(t32, out32, dv.stream_of(dev)) is False and out32.dtype == torch.float32
            assert np.array_equal(out32.cpu().numpy(), np.interp(q32.astype('f8'), x, f).astype('f4')), name
        # outside / NaN: NaN there, the flag raised (and lowered again for the next call)
        bad = torch.as_tensor(np.concatenate([q[:1000], [np.nextafter(x[0], -np.inf), np.nextafter(x[-1], np.inf), np.nan]]), device=dev)
        out = torch.empty_like(bad)
        assert table(bad, out, dv.stream_of(dev)) is True, name
        got = out.cpu().numpy()
        assert np.isnan(got[-3:]).all() and np.array_equal(got[:-3], np.interp(q[:1000], x, f)), name
        assert table(tq[:10], torch.empty(10, dtype=torch.float64, device=dev), dv.stream_of(dev)) is False, name
    import copy
    clone = copy.deepcopy(table)
    assert clone.handle.value != table.handle.value and clone.law == table.law


def test_spline_points_of_a_catalogue():
    """``cp_spline_points`` on a catalogue (one spline, >= 65 536 points: knots and interval coefficients in LDS) against the same call on pieces short
    enough to take the general kernel (coefficients formed per sample): the same bits, for values and derivatives, with and without extrapolation; and
    against scipy's CubicSpline."""
    import torch
    from scipy.interpolate import CubicSpline
    from cosmoprimo_amd import _lib, _device as dv
    from cosmoprimo_amd.spline import dense_operator
    dev = torch.device('cuda:0')
    lib, stream = _lib.load(), dv.stream_of(dev)
    rng = np.random.default_rng(5)
    for n in [2, 3, 119, 400, 2048, 2049]:
        xk = np.sort(np.concatenate([[0.], rng.uniform(0., 10., n - 2), [10.]])) if n > 2 else np.array([0., 10.])
        y = np.sin(xk) + 0.1 * xk**2
        slopes = dense_operator(xk, xk, bc='natural', nu=1).dot(y) if n > 2 else np.full(2, (y[1] - y[0]) / 10.)
        xq = np.concatenate([rng.uniform(-1., 11., 200000), xk, [np.nan]])
        txk, ty, ts, txq = [torch.as_tensor(a, device=dev) for a in (xk, y, slopes, xq)]
        for nu in (0, 1, 2):
            for extrapolate in (0, 1):
                whole = torch.empty_like(txq)
                _lib.check(lib.cp_spline_points(txk.data_ptr(), ty.data_ptr(), ts.data_ptr(), n, 1, txq.data_ptr(), whole.data_ptr(), txq.numel(), nu, extrapolate,
                                                0, stream))
                pieces = torch.empty_like(txq)
                for lo in range(0, txq.numel(), 50000):
                    m = min(50000, txq.numel() - lo)
                    _lib.check(lib.cp_spline_points(txk.data_ptr(), ty.data_ptr(), ts.data_ptr(), n, 1, txq.data_ptr() + 8 * lo, pieces.data_ptr() + 8 * lo, m, nu,
                                                    extrapolate, 0, stream))
                got, ref = whole.cpu().numpy(), pieces.cpu().numpy()
                assert np.array_equal(got, ref, equal_nan=True), (n, nu, extrapolate)
                assert np.isnan(got[-1]) and (extrapolate or np.isnan(got[:200000][(xq[:200000] < 0.) | (xq[:200000] > 10.)]).all())
        if n > 2:
            inside = (xq >= 0.) & (xq <= 10.)
            _lib.check(lib.cp_spline_points(txk.data_ptr(), ty.data_ptr(), ts.data_ptr(), n, 1, txq.data_ptr(), whole.data_ptr(), txq.numel(), 0, 0, 0, stream))
            np.testing.assert_allclose(whole.cpu().numpy()[inside], CubicSpline(xk, y, bc_type='natural')(xq[inside]), rtol=1e-11, atol=1e-12)
