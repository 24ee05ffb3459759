"""Batches of tabulated P(k, z) (BASELINE config 3, variant B of SURVEY.md 8(d): one 500 x 30 table per cosmology): every entry of the batch
must be what the single-table interpolator gives for that table -- whose numbers are pinned by the reference's (tests/golden/sigma.npz) --
for evaluation on grids and at pairs, sigma_rz, sigma8_z, sigma_dz and the sigma8 rescaling; a table with a negative entry stays NaN alone."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def cp():
    import torch
    assert torch.cuda.is_available(), 'GPU tests need a ROCm device'
    import cosmoprimo_amd
    return cosmoprimo_amd


def tables(g, nb, seed=0):
    rng = np.random.default_rng(seed)
    k, z, pk = g['table_k'], g['table_z'], g['table_pk']
    amp, tilt, evol = rng.uniform(0.5, 2., nb), rng.uniform(-0.1, 0.1, nb), rng.uniform(-0.2, 0.2, nb)
    amp[0], tilt[0], evol[0] = 1., 0., 0.                       # entry 0 is the golden table itself
    batch = amp[:, None, None] * (k[None, :, None] / 0.05)**tilt[:, None, None] * (1. + z[None, None, :])**evol[:, None, None] * pk[None]
    return k, z, batch


def test_batch_of_tables_matches_single_tables(cp, golden):
    import torch
    g = golden('sigma')
    k, z, batch = tables(g, 7)
    r, zq = g['r'], g['z']
    many = cp.PowerSpectrumInterpolator2D(k, z, torch.as_tensor(batch, device='cuda'))
    out = many.sigma_rz(r, zq)
    assert out.shape == (7, 256, 64)
    np.testing.assert_allclose(out[0], g['table_sigma_rz'], rtol=1e-9)              # the reference's numbers for its table
    from oracle import interp as oi, sigma as osig
    for i in range(1, 7):      # ... and the oracle's restatement of the reference's route for the perturbed ones
        pk2d = oi.pk_interp_2d(k, z, batch[i])
        np.testing.assert_allclose(out[i], np.sqrt(osig.sigma_r2(r, lambda kk: pk2d(kk, zq, grid=True))), rtol=1e-9, err_msg='table %d' % i)
    kq, zz = g['table_eval_k'], np.linspace(0.1, 2.5, 16)
    np.testing.assert_allclose(many(kq, zq[::4])[0], g['table_eval'], rtol=1e-10)               # ... and its values (64 k x 16 z)
    pairs = many(kq[:16], zz, grid=False)
    assert pairs.shape == (7, 16)
    for i in range(7):
        one = cp.PowerSpectrumInterpolator2D(k, z, batch[i])
        np.testing.assert_allclose(out[i], one.sigma_rz(r, zq), rtol=1e-11)
        np.testing.assert_allclose(many(kq, zz)[i], one(kq, zz), rtol=1e-12)
        np.testing.assert_allclose(pairs[i], one(kq[:16], zz, grid=False), rtol=1e-12)
        np.testing.assert_allclose(many.sigma8_z(zq[::9])[i], one.sigma8_z(zq[::9]), rtol=1e-11)
        np.testing.assert_allclose(many.sigma_dz(zq[::9])[i], one.sigma_dz(zq[::9]), rtol=1e-11)
    # outside the redshift range: NaN, and nowhere else
    high = many.sigma_rz(r[:4], np.array([1., 3.5]))
    assert np.isnan(high[..., 1]).all() and np.isfinite(high[..., 0]).all()
    # grids given in any order
    shuffled = cp.PowerSpectrumInterpolator2D(k[::-1], z[::-1], batch[:, ::-1, ::-1].copy())
    np.testing.assert_allclose(shuffled(kq, zz), many(kq, zz), rtol=1e-14)
    # one table with a negative entry: that cosmology is NaN, the others are what they were
    bad = batch.copy()
    bad[3, 100, 7] *= -1.
    mixed = cp.PowerSpectrumInterpolator2D(k, z, bad)
    res = mixed.sigma_rz(r[::16], zq[::8])
    assert np.isnan(res[3]).all() and np.isfinite(np.delete(res, 3, axis=0)).all()
    np.testing.assert_allclose(np.delete(res, 3, axis=0), np.delete(out[:, ::16, ::8], 3, axis=0), rtol=1e-12)
    # sigma8 rescaling, one factor per table
    many.rescale_sigma8(0.8)
    np.testing.assert_allclose(many.sigma8_z(0.), 0.8, rtol=1e-10)
    assert many.pk.shape == (7, 500, 30)


@pytest.mark.parametrize('nk,nzq', [(1024, 64), (300, 17), (1000, 64), (257, 1), (4100, 33)])
def test_rows_of_tables_both_routes(cp, golden, nk, nzq):
    """Interpolator2D.rows_y_major for a batch of tables (the rows of P(k, z) the sigma integrals transform): the kernel that evaluates the k splines from
    the tables' second derivatives and contracts z on the matrix cores (cp_tables_rows_direct: whole tiles through its table-driven 10^x, partial
    tiles, NaN tables and few redshifts through the general epilogue) against the two banded operators, for sizes that do and do not fill its tiles."""
    import torch
    from cosmoprimo_amd import interpolator as itp
    g = golden('sigma')
    k, z, batch = tables(g, 5, seed=nk)
    batch[2, 40, 3] = -1.      # (the logarithm of this table is NaN: the whole surface)
    many = cp.PowerSpectrumInterpolator2D(k, z, torch.as_tensor(batch, device='cuda'))
    # (away from the two ends of the tables: _pad_log puts its extrapolation knots within 1e-9 of the end knots when the table reaches the extrapolation
    # range, as this one does at k = 100 -- the last intervals of such a spline are conditioned like 1e9, and the two routes, like FITPACK, differ by 1e-7 there)
    kq = np.geomspace(k[16], k[-17], nk)      # (the disturbance decays by 0.27 per knot)
    zq = np.linspace(z[0], z[-1], nzq) if nzq > 1 else np.array([0.7])
    saved = itp._DIRECT_K_SPLINE, itp._PAIRED_TABLES
    try:
        itp._DIRECT_K_SPLINE = True
        direct = many._interp.rows_y_major(kq, zq, exp10=True).cpu().numpy()      # (tables and second derivatives as pairs in one array)
        assert many._interp._fun_y_major_m.shape == many._interp._fun_y_major.shape + (2,)
        itp._PAIRED_TABLES = False
        del many._interp._fun_y_major_m
        two_arrays = many._interp.rows_y_major(kq, zq, exp10=True).cpu().numpy()
        assert many._interp._fun_y_major_m.shape == many._interp._fun_y_major.shape
        itp._DIRECT_K_SPLINE = False
        operators = many._interp.rows_y_major(kq, zq, exp10=True).cpu().numpy()
    finally:
        itp._DIRECT_K_SPLINE, itp._PAIRED_TABLES = saved
    assert direct.shape == (5, nzq, nk)
    assert np.array_equal(direct, two_arrays, equal_nan=True)      # the same numbers in the same operations, read from another layout
    assert np.isnan(direct[2]).all() and np.isnan(operators[2]).all()
    keep = [0, 1, 3, 4]
    np.testing.assert_allclose(direct[keep], operators[keep], rtol=1e-11)
    one = cp.PowerSpectrumInterpolator2D(k, z, batch[4])
    np.testing.assert_allclose(direct[4], one(kq, zq).T, rtol=1e-11)


def test_config3_variant_b_full_size(cp, golden):
    """10 000 tables (500 k x 30 z) -> sigma_rz on 256 r x 64 z: finite, deterministic, falling with r and z, sampled entries against single tables."""
    import torch
    g = golden('sigma')
    k, z, batch = tables(g, 10000, seed=1)
    r, zq = torch.as_tensor(g['r'], device='cuda'), torch.as_tensor(g['z'], device='cuda')
    many = cp.PowerSpectrumInterpolator2D(k, z, torch.as_tensor(batch, device='cuda'))
    out = many.sigma_rz(r, zq)
    assert tuple(out.shape) == (10000, 256, 64) and bool(torch.isfinite(out).all())
    assert torch.equal(out, many.sigma_rz(r, zq))
    assert bool((out[:, 1:] < out[:, :-1]).all())
    for i in (0, 5000, 9999):
        one = cp.PowerSpectrumInterpolator2D(k, z, batch[i])
        np.testing.assert_allclose(out[i].cpu().numpy(), one.sigma_rz(g['r'], g['z']), rtol=1e-11)
    # sampled PERTURBED tables against the oracle's restatement of the reference's route (RectBivariateSpline of log10 P on (log10 k, z) with the
    # log-log padding, P(k, z) at the 64 redshifts, one TophatVariance FFTLog per redshift, natural spline to r: interpolator.py:846-875) -- nothing
    # of the package on the reference side
    from oracle import interp as oi, sigma as osig
    for i in (1, 4242, 9998):
        pk2d = oi.pk_interp_2d(k, z, batch[i])
        ref = np.sqrt(osig.sigma_r2(g['r'], lambda kk: pk2d(kk, g['z'], grid=True)))      # (nr, nz)
        np.testing.assert_allclose(out[i].cpu().numpy(), ref, rtol=1e-9, err_msg='table %d' % i)


def test_quad_method(cp, golden):
    """method='quad' of the sigma integrals: the reference hands epsabs = epsrel = 1e-5 to scipy.integrate.quad per (r, column)
    (interpolator.py:167-177, 255-273); here a composite Simpson rule refined for the whole batch until the same tolerances are met.
    Both are then within 1e-5 of the integral: 2e-5 between them."""
    g = golden('sigma_quad')
    k, z, pk, r = g['table_k'], g['table_z'], g['table_pk'], g['r']
    one = cp.PowerSpectrumInterpolator1D(k, pk[:, 0])
    np.testing.assert_allclose(one.sigma_r(r, method='quad'), g['sigma_r'], rtol=2e-5)
    np.testing.assert_allclose(one.sigma_d(method='quad'), g['sigma_d'], rtol=2e-5)
    # asked for 1e-10, the reference's quad stops at its 50 subintervals, 2e-8 short; the same integrand given room converges to our value
    tight = one.sigma_r(r, method='quad', epsabs=1e-10, epsrel=1e-10)
    np.testing.assert_allclose(tight, g['sigma_r_tight'], rtol=1e-7)
    np.testing.assert_allclose(tight, g['sigma_r_converged'], rtol=1e-10)
    np.testing.assert_allclose(one.sigma_r(r, method='quad', epsabs=1e-10, epsrel=1e-10), one.sigma_r(r, method='simpson', nk=2**15 + 1), rtol=1e-9)
    # the 2-D classes (the reference's own fail with this method; its 1-D class per redshift is the expected value)
    two = cp.PowerSpectrumInterpolator2D(k, z, pk)
    np.testing.assert_allclose(two.sigma_rz(r, g['z'], method='quad'), g['sigma_rz'], rtol=2e-5)
    np.testing.assert_allclose(two.sigma_dz(g['z'], method='quad'), g['sigma_dz'], rtol=2e-5)
    with pytest.raises(NotImplementedError):
        one.sigma_r(r, method='romberg')
