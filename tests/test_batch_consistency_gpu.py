"""A batch of cosmologies (array-valued parameters) against the same cosmologies one at a time, across the API: the batch axis leads the results
of the sections and of the 2D interpolators, and becomes the trailing columns of 1D interpolators (``to_1d``, ``Primordial.pk_interpolator``)
and of what is built on them (filters, ``to_xi``).  What a batch cannot do says so."""
import warnings

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def same(batched, singles, rtol=1e-10, cols=False, atol=0.):
    batched = np.asarray(batched)
    if cols:
        batched = np.moveaxis(batched, 1, 0)
    assert batched.shape[0] == len(singles)
    for i, single in enumerate(singles):
        np.testing.assert_allclose(batched[i], np.asarray(single), rtol=rtol, atol=atol * np.nanmax(np.abs(single)), equal_nan=True)


def test_batch_consistency():
    import cosmoprimo_amd as cp
    from cosmoprimo_amd.interpolator import _host
    warnings.simplefilter('ignore')
    Om = np.array([0.28, 0.30, 0.33])
    batch = cp.Cosmology(engine='eisenstein_hu', Omega_m=Om)
    singles = [cp.Cosmology(engine='eisenstein_hu', Omega_m=float(o)) for o in Om]
    k, z, r, s = np.geomspace(1e-3, 1., 50), np.array([0., 0.5, 1.]), np.array([2., 8., 30.]), np.geomspace(1., 150., 40)
    pkb, pks = batch.get_fourier().pk_interpolator(), [c.get_fourier().pk_interpolator() for c in singles]
    same(pkb(k, z), [p(k, z) for p in pks])
    same(pkb.sigma_rz(r, z), [p.sigma_rz(r, z) for p in pks])
    same(pkb.sigma_dz(z), [p.sigma_dz(z) for p in pks])
    same(pkb.sigma8_z(z), [p.sigma8_z(z) for p in pks])
    same(pkb.growth_rate_rz(r, z), [p.growth_rate_rz(r, z) for p in pks], 1e-7)
    # 1D interpolators: the cosmologies are columns
    p1b, p1s = pkb.to_1d(z=0.5), [p.to_1d(z=0.5) for p in pks]
    assert p1b(k).shape == (50, 3) and pkb.to_1d(z=z)(k).shape == (50, 3, 3)
    same(p1b(k), [p(k) for p in p1s], cols=True)
    same(p1b.sigma_r(r), [p.sigma_r(r) for p in p1s], cols=True)
    same(p1b.to_xi()(s), [p.to_xi()(s) for p in p1s], 1e-8, cols=True)
    same(p1b.clone(extrap_kmin=1e-5).to_xi().to_pk()(k), [p.clone(extrap_kmin=1e-5).to_xi().to_pk()(k) for p in p1s], 1e-7, cols=True)
    # filters that act column by column
    for engine in ['wallish2018', 'savgol', 'ehsavgol']:
        fb = cp.PowerSpectrumBAOFilter(p1b, engine=engine)
        assert fb.pknow.shape == (1024, 3)
        same(fb.pknow, [cp.PowerSpectrumBAOFilter(p, engine=engine).pknow for p in p1s], 1e-7, cols=True)
    fb = cp.PowerSpectrumBAOFilter(pkb, engine='wallish2018')                 # 2D interpolator of the batch: (batch, nk, nz)
    same(fb.pknow, [cp.PowerSpectrumBAOFilter(p, engine='wallish2018').pknow for p in pks], 1e-7)
    # the filters whose operator depends on the cosmology (its rs_drag ratio, its no-wiggle template): one per column of the batch
    fid = cp.Cosmology(engine='eisenstein_hu')
    for engine in ['ehpoly', 'peakaverage', 'ehsavgol', 'hinton2017']:
        fb = cp.PowerSpectrumBAOFilter(p1b, engine=engine, cosmo=batch, cosmo_fid=fid)
        assert fb.pknow.shape == (1024, 3)
        same(fb.pknow, [cp.PowerSpectrumBAOFilter(p, engine=engine, cosmo=c, cosmo_fid=fid).pknow for p, c in zip(p1s, singles)], 1e-9 if engine != 'hinton2017' else 1e-7,
             cols=True)
    fb = cp.PowerSpectrumBAOFilter(pkb, engine='wallish2018')
    fs = [cp.PowerSpectrumBAOFilter(p, engine='wallish2018') for p in pks]
    # to_xi of the batch: ONE transform per cosmology, the growth factor handed on; (batch, ns, nz)
    xib, xis = pkb.to_xi(), [p.to_xi() for p in pks]
    assert xib(s, z).shape == (3, 40, 3) and xib(s[:3], z, grid=False).shape == (3, 3)
    same(xib(s, z), [x(s, z) for x in xis], 1e-8, atol=1e-13)
    same(xib(s[:3], z, grid=False), [x(s[:3], z, grid=False) for x in xis], 1e-8, atol=1e-13)
    same(xib(s, z, ignore_growth=True), [x(s, z, ignore_growth=True) for x in xis], 1e-8, atol=1e-13)
    assert np.isnan(xib(s, [-0.5, 0.5])[:, :, 0]).all() and not np.isnan(xib(s, [-0.5, 0.5])[:, :, 1]).any()
    assert xib.to_1d(z=0.5)(s).shape == (40, 3) and xib.to_1d(z=z)(s).shape == (40, 3, 3)
    same(xib.to_1d(z=0.5)(s), [x.to_1d(z=0.5)(s) for x in xis], 1e-8, cols=True, atol=1e-13)
    same(xib.to_1d(z=0.5).to_pk()(k[5:45]), [x.to_1d(z=0.5).to_pk()(k[5:45]) for x in xis], 1e-6, cols=True)
    # ... and back: a batch of (k, z) tables with the growth factor
    back, backs = pkb.clone(extrap_kmin=1e-5).to_xi().to_pk(), [p.clone(extrap_kmin=1e-5).to_xi().to_pk() for p in pks]
    same(back(k, z), [b(k, z) for b in backs], 1e-7)
    same(back.sigma8_z(z), [b.sigma8_z(z) for b in backs], 1e-7)
    same(xib.sigma8_z(z), [x.sigma8_z(z) for x in xis], 1e-7)
    # the smooth interpolators of a filter of the batch: tables of the batch with the input's growth factor
    smooth, smooths = fb.smooth_pk_interpolator(), [f.smooth_pk_interpolator() for f in fs]
    same(smooth(k, z), [sm(k, z) for sm in smooths], 1e-7)
    same(fb.smooth_xi_interpolator()(s, z), [f.smooth_xi_interpolator()(s, z) for f in fs], 1e-6, atol=1e-9)
    with pytest.raises(NotImplementedError):
        batch.solve('h', 'theta_MC_100', 1.04)
    s_arr, z_arr, xi_arr = pkb.to_xi_arrays()
    assert xi_arr.shape == (3, s_arr.size, z_arr.size)
    same(xi_arr, [p.to_xi_arrays()[2] for p in pks], 1e-8, atol=1e-13)      # two rows share a transform: rounding is relative to the larger one
    # sections
    ba = batch.get_background()
    for name in ['efunc', 'comoving_radial_distance', 'time', 'growth_factor', 'growth_rate', 'rho_m', 'Omega_m', 'T_cmb', 'luminosity_distance']:
        same(getattr(ba, name)(z), [getattr(c.get_background(), name)(z) for c in singles])
    same(ba.age, [c.get_background().age for c in singles])
    same(_host(batch.get_thermodynamics().rs_drag), [c.get_thermodynamics().rs_drag for c in singles])
    same(batch['theta_cosmomc'], [c['theta_cosmomc'] for c in singles])
    nu = cp.Cosmology(engine='eisenstein_hu_nowiggle_variants', m_ncdm=[0.06], Omega_m=Om)
    nus = [cp.Cosmology(engine='eisenstein_hu_nowiggle_variants', m_ncdm=[0.06], Omega_m=float(o)) for o in Om]
    same(nu.get_fourier().pk_interpolator()(k, z), [c.get_fourier().pk_interpolator()(k, z) for c in nus])
    same(nu.get_fourier().pk_interpolator(of='delta_cb')(k, z), [c.get_fourier().pk_interpolator(of='delta_cb')(k, z) for c in nus])
    same(nu.get_fourier().sigma8_z(z), [c.get_fourier().sigma8_z(z) for c in nus])


def test_batches_of_tables_with_a_growth_factor():
    """(batch, nk, nz) tables times growth_factor_sq(z), and (batch, nk, 1) columns times it, against the same tables one at a time."""
    import cosmoprimo_amd as cp
    warnings.simplefilter('ignore')
    g = np.load(__import__('os').path.join(__import__('os').path.dirname(__file__), 'golden', 'sigma.npz'))
    kt, zt, table = g['table_k'], g['table_z'], g['table_pk']
    amp = np.array([0.7, 1., 1.6])
    tables = amp[:, None, None] * table[None]
    k, z, r = np.geomspace(2e-4, 5., 60), np.array([0.1, 0.7, 1.4]), np.array([2., 8., 30.])

    def growth_one(zz):
        return 1. / (1. + np.asarray(zz))**2

    def growth_each(zz):
        return amp[:, None]**0.5 / (1. + np.asarray(zz))**2

    for growth, singles_growth in [(growth_one, [growth_one] * 3), (growth_each, [lambda zz, a=a: a**0.5 / (1. + np.asarray(zz))**2 for a in amp])]:
        b = cp.PowerSpectrumInterpolator2D(kt, zt, tables, growth_factor_sq=growth)
        ones = [cp.PowerSpectrumInterpolator2D(kt, zt, t, growth_factor_sq=gs) for t, gs in zip(tables, singles_growth)]
        same(b(k, z), [o(k, z) for o in ones])
        same(b(k[:3], z, grid=False), [o(k[:3], z, grid=False) for o in ones])
        same(b(k, z, ignore_growth=True), [o(k, z, ignore_growth=True) for o in ones])
        same(b.sigma_rz(r, z), [o.sigma_rz(r, z) for o in ones], 1e-9)
        same(b.sigma_dz(z), [o.sigma_dz(z) for o in ones], 1e-9)
        # one column per cosmology times the growth factor
        b1 = cp.PowerSpectrumInterpolator2D(kt, zt[:1], tables[:, :, :1], growth_factor_sq=growth)
        ones1 = [cp.PowerSpectrumInterpolator2D(kt, zt[:1], t[:, :1], growth_factor_sq=gs) for t, gs in zip(tables, singles_growth)]
        same(b1(k, z), [o(k, z) for o in ones1])
        same(b1(k[:3], z, grid=False), [o(k[:3], z, grid=False) for o in ones1])
        same(b1.sigma_rz(r, z), [o.sigma_rz(r, z) for o in ones1], 1e-9)
        same(b1.to_xi()(r, z), [o.to_xi()(r, z) for o in ones1], 1e-8, atol=1e-13)
    # to_xi of a batch of tables: one (s, z) surface per table
    b = cp.PowerSpectrumInterpolator2D(kt, zt, tables)
    ones = [cp.PowerSpectrumInterpolator2D(kt, zt, t) for t in tables]
    s = np.geomspace(1., 150., 30)
    same(b.to_xi()(s, z), [o.to_xi()(s, z) for o in ones], 1e-8, atol=1e-13)
    same(b.to_xi()(s[:3], z, grid=False), [o.to_xi()(s[:3], z, grid=False) for o in ones], 1e-8, atol=1e-13)
    same(b.to_xi().to_pk()(k[10:40], z), [o.to_xi().to_pk()(k[10:40], z) for o in ones], 1e-7)
