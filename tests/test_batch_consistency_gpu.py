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
    with pytest.raises(NotImplementedError):
        pkb.to_xi()
    with pytest.raises(NotImplementedError):
        fb.smooth_pk_interpolator()
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
