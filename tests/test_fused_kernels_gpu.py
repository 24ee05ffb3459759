"""GPU: the round-3 fused entry points called directly through the C ABI, against the separate kernels / plain torch arithmetic they replace:
cp_tables_rows (both spline directions of a batch of tables), cp_fftlog_spline_execute (FFTLog + spline to radii), the elementwise passes of
the two BAO filters (cp_wallish_finish, cp_brieden_ratio / _knots / _finish).  Odd sizes, NaN rows and queries, partial tiles."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _env():
    import torch
    from cosmoprimo_amd import _lib, _device as dv
    dev = torch.device('cuda', 0)
    return torch, _lib, _lib.load(), dv, dev


@pytest.mark.parametrize('shape', [(7, 30, 504, 64, 1024), (3, 17, 200, 50, 1000), (1, 32, 64, 64, 64), (5, 4, 40, 9, 130)])
@pytest.mark.parametrize('post', [None, 'exp10', 'sqrt'])
def test_tables_rows_matches_two_launches(shape, post):
    torch, _lib, lib, dv, dev = _env()
    from cosmoprimo_amd.spline import LinearOperator, dense_operator
    nb, nzin, n, nzq, nq = shape
    rng = np.random.default_rng(nb + n)
    x = np.linspace(0., 10., n) + rng.uniform(-0.3, 0.3, n) * 10. / n                        # uneven knots, no near-coincident ones (bounded splines)
    xq = np.sort(rng.uniform(-0.5, 10.5, nq))                                                # some queries outside the knots: NaN rows of the operator
    zk, zq = np.linspace(0., 3., nzin), np.linspace(0., 3., nzq)
    opx = LinearOperator.spline(x, xq, bc='not-a-knot', extrapolate=False, device=dev)
    opz = LinearOperator.dense(dense_operator(zk, zq, bc='not-a-knot', extrapolate=True), device=dev)
    assert lib.cp_tables_rows_available(opx._handle, opz._handle) == 1
    smooth = 1. + 0.3 * np.sin(x)[None, None, :] * np.cos(zk)[None, :, None] * rng.uniform(0.5, 1., (nb, 1, 1))
    t = torch.as_tensor(smooth + 1e-3 * rng.standard_normal((nb, nzin, n)), device=dev)
    out = torch.full((nb, nzq, nq), -7., dtype=torch.float64, device=dev)
    code = {None: 0, 'sqrt': 1, 'exp10': 2}[post]
    _lib.check(lib.cp_tables_rows(opx._handle, opz._handle, t.data_ptr(), out.data_ptr(), nb, code, 1., dv.stream_of(dev)))
    ref = opz.mid(opx(t), post=post)
    got, ref = out.cpu().numpy(), ref.cpu().numpy()
    outside = (xq < x[0]) | (xq > x[-1])
    assert np.isnan(got[..., outside]).all() and np.isfinite(got[..., ~outside]).all()
    np.testing.assert_allclose(got[..., ~outside], ref[..., ~outside], rtol=1e-12, atol=1e-13)
    # and against scipy on one table: the reference's own two passes (jax.py:241-271)
    from scipy.interpolate import CubicSpline
    inside = ~outside
    one = CubicSpline(x, t[0].cpu().numpy(), axis=1)(xq[inside])                              # (nzin, nq_in)
    two = CubicSpline(zk, one, axis=0)(zq) if nzin >= 4 else None
    if two is not None:
        two = two if post is None else (np.sqrt(two) if post == 'sqrt' else 10.**two)
        np.testing.assert_allclose(got[0][:, inside], two, rtol=1e-10, atol=1e-12)
    # a table holding a NaN is NaN throughout, its neighbours are untouched
    if nb > 1:
        t2 = t.clone()
        t2[1, nzin // 2, n // 3] = float('nan')
        _lib.check(lib.cp_tables_rows(opx._handle, opz._handle, t2.data_ptr(), out.data_ptr(), nb, code, 1., dv.stream_of(dev)))
        again = out.cpu().numpy()
        assert np.isnan(again[1]).any() and np.array_equal(np.delete(again, 1, axis=0), np.delete(got, 1, axis=0), equal_nan=True)


def test_tables_rows_refuses_other_shapes():
    torch, _lib, lib, dv, dev = _env()
    from cosmoprimo_amd.spline import LinearOperator
    rng = np.random.default_rng(0)
    opx = LinearOperator.spline(np.linspace(0., 1., 50), np.linspace(0., 1., 80), device=dev)
    wide = LinearOperator.dense(rng.normal(size=(70, 40)), device=dev)          # more than 64 output redshifts, more than 32 input ones
    assert lib.cp_tables_rows_available(opx._handle, wide._handle) == 0
    t = torch.zeros((2, 40, 50), dtype=torch.float64, device=dev)
    out = torch.zeros((2, 70, 80), dtype=torch.float64, device=dev)
    assert lib.cp_tables_rows(opx._handle, wide._handle, t.data_ptr(), out.data_ptr(), 2, 0, 1., dv.stream_of(dev)) != 0


@pytest.mark.parametrize('nrows,nq', [(1, 1), (2, 40), (33, 256), (300, 130)])
def test_fftlog_spline_execute(nrows, nq):
    torch, _lib, lib, dv, dev = _env()
    import cosmoprimo_amd as cp
    from cosmoprimo_amd.spline import LinearOperator
    k = np.geomspace(1e-7, 1e2, 1024)
    fft = cp.TophatVariance(k, device=dev)
    r = np.geomspace(0.7, 150., nq) if nq > 1 else np.array([8.])
    if nq > 4:
        r[0], r[-1] = 1e-5, 1e9                                                       # outside the output grid of the transform: NaN
    op = LinearOperator.spline(fft.y[0], r, bc='natural', device=dev)
    plan = fft._get_plan(dev)
    assert lib.cp_sigma_rz_fused_available(plan.handle, op._handle) == 1
    rng = np.random.default_rng(nrows)
    rows = torch.as_tensor(rng.uniform(0.5, 2., (nrows, 1)) * (k / 0.05)**rng.uniform(-2.2, -1.8, (nrows, 1)) * 1e4, device=dev).contiguous()
    if nrows > 2:
        rows[1, 100] = float('nan')                                                   # row 1 shares a transform with row 0
    for post in (0, 1):
        out = torch.full((nrows, nq), -3., dtype=torch.float64, device=dev)
        _lib.check(lib.cp_fftlog_spline_execute(plan.handle, op._handle, rows.data_ptr(), out.data_ptr(), nrows, post, dv.stream_of(dev)))
        ref = op(fft(rows)[1], sqrt=bool(post)).cpu().numpy()
        got = out.cpu().numpy()
        assert np.array_equal(np.isnan(got), np.isnan(ref))
        # (the separate spline kernel may take the matrix cores: another order of additions, 2e-15 of the ROW's largest value, which is 1e3 times
        # the variances at the largest radii)
        np.testing.assert_allclose(got[np.isfinite(ref)], ref[np.isfinite(ref)], rtol=2e-11)
        if nrows > 2:
            assert np.isnan(got[1]).all() and np.isfinite(got[0][1:-1] if nq > 4 else got[0]).all()
    # a transform of another size is refused (the caller makes the two calls)
    other = cp.TophatVariance(np.geomspace(1e-5, 1e2, 512), device=dev)
    op2 = LinearOperator.spline(other.y[0], np.array([8.]), bc='natural', device=dev)
    assert lib.cp_sigma_rz_fused_available(other._get_plan(dev).handle, op2._handle) == 0


@pytest.mark.parametrize('nrows,nq,group', [(1, 1, 0), (2, 40, 0), (33, 256, 0), (301, 130, 0), (64, 256, 64), (6 * 10, 100, 10), (9 * 64, 256, 64), (4 * 2, 511, 2)])
@pytest.mark.parametrize('prefiltered', [True, False])
def test_fftlog_geospline_execute(nrows, nq, group, prefiltered, monkeypatch):
    """cp_fftlog_geospline_execute (FFTLog + natural spline on the geometric output grid, SOLVED on the CU by two first-order recursions, or --
    prefiltered -- evaluated from B-spline coefficients that the transform itself delivers, the solve being a division of its u) against scipy's
    CubicSpline(bc_type='natural') of the package's own transformed rows (1e-13 in tilted space) and against the band-operator route; plain and grouped (transposed) layouts, odd batches, a NaN row next to good ones, queries outside the grid, tables
    (groups) that do not fill the eight XCD shares."""
    from scipy.interpolate import CubicSpline
    torch, _lib, lib, dv, dev = _env()
    import cosmoprimo_amd as cp
    from cosmoprimo_amd import interpolator as itp
    from cosmoprimo_amd.spline import LinearOperator
    monkeypatch.setattr(itp, '_GEOSPLINE_PREFILTERED', prefiltered)
    k = np.geomspace(1e-7, 1e2, 1024)
    fft = cp.TophatVariance(k, device=dev)
    s = fft.y[0]
    r = np.geomspace(0.7, 150., nq) if nq > 1 else np.array([8.])
    if nq > 4:
        r[0], r[-1] = 1e-5, 1e9                                                       # outside the output grid of the transform: NaN
    rng = np.random.default_rng(nrows + nq)
    rows = torch.as_tensor(rng.uniform(0.5, 2., (nrows, 1)) * (k / 0.05)**rng.uniform(-2.2, -1.8, (nrows, 1)) * 1e4, device=dev).contiguous()
    if nrows > 2:
        rows[1, 100] = float('nan')                                                   # row 1 shares a transform with row 0
    var = fft(rows)[1].cpu().numpy()
    inside = (r >= s[0]) & (r <= s[-1])
    op = LinearOperator.spline(s, r, bc='natural', device=dev)
    for sqrt in (False, True):
        shaped = rows.reshape(nrows // group, group, 1024) if group else rows
        got = itp._fftlog_then_geospline(fft, s, r, shaped, dev, sqrt=sqrt, group=group)
        assert got is not None
        got = got.cpu().numpy()
        if group:
            assert got.shape == (nrows // group, nq, group)
            got = got.transpose(0, 2, 1).reshape(nrows, nq)
        assert got.shape == (nrows, nq)
        assert np.isnan(got[:, ~inside]).all()
        for i in range(nrows):
            if nrows > 2 and i == 1:
                assert np.isnan(got[i]).all()
                continue
            # the reference splines the rows ANOTHER launch transformed: two evaluations of an FFTLog agree to ~1e-14 in tilted space (g y^1.5
            # against its largest value: tests/test_fftlog_gpu.py), i.e. worse by (r_max / r)^1.5 at the small radii; the solve itself adds 1e-15
            ref = CubicSpline(s, var[i], bc_type='natural')(r[inside])
            tilted = np.abs((got[i, inside]**2 if sqrt else got[i, inside]) - ref) * r[inside]**1.5
            assert tilted.max() < 1e-13 * np.abs(var[i] * s**1.5).max(), 'row %d' % i
        plan = [v for key, v in itp._op_cache.items() if key[0] == 'geospline' and key[2] == prefiltered][-1]
        assert plan.prefiltered == prefiltered
        # the kernel with the same front end and the band operator behind it transforms with the same arithmetic: what is left is the solve
        same_front = itp._fftlog_then_spline(fft, op, rows, dev, sqrt=sqrt)
        banded = op(torch.as_tensor(var, device=dev), sqrt=sqrt).cpu().numpy()
        if same_front is not None:
            same_front = same_front.cpu().numpy()
            assert np.array_equal(np.isfinite(got), np.isfinite(same_front))
        if prefiltered:      # another u: another evaluation of the transform, compared as the reference above is
            for other in (same_front, banded):
                if other is None:
                    continue
                for i in range(nrows):
                    if nrows > 2 and i == 1:
                        continue
                    a, b = (x[i, inside]**2 if sqrt else x[i, inside] for x in (got, other))
                    assert (np.abs(a - b) * r[inside]**1.5).max() < 1e-13 * np.abs(var[i] * s**1.5).max(), 'row %d' % i
            continue
        if same_front is not None:
            keep = np.isfinite(same_front)
            np.testing.assert_allclose(got[keep], same_front[keep], rtol=1e-12)
        keep = np.isfinite(banded)
        np.testing.assert_allclose(got[keep], banded[keep], rtol=2e-11)


@pytest.mark.parametrize('nrows,nq,group', [(20001, 256, 0), (16384, 300, 0), (12 * 1100, 256, 12), (64 * 330, 77, 64)])
def test_prefiltered_geospline_over_many_pairs_per_workgroup(nrows, nq, group):
    """Batches of several pairs per workgroup (the evaluation of a pair is deferred into the next one's first phase; the last pair of a workgroup is evaluated
    behind its loop): EVERY row of the prefiltered form against the form that solves the spline on the CU from the ordinary transform -- two evaluations of an
    FFTLog with different u, compared in tilted space --, odd batches, more than 256 radii (eight blocks of radii per workgroup), grouped layouts whose
    tables do not fill the XCD shares evenly."""
    torch, _lib, lib, dv, dev = _env()
    import cosmoprimo_amd as cp
    from cosmoprimo_amd import interpolator as itp
    k = np.geomspace(1e-7, 1e2, 1024)
    fft = cp.TophatVariance(k, device=dev)
    s = fft.y[0]
    r = np.geomspace(0.8, 120., nq)
    rng = np.random.default_rng(nrows + nq)
    amp, tilt = (torch.as_tensor(x, device=dev) for x in (rng.uniform(0.5, 2., (nrows, 1)), rng.uniform(-2.2, -1.8, (nrows, 1))))
    rows = (amp * 1e4 * torch.as_tensor(k / 0.05, device=dev)[None, :]**tilt).contiguous()
    shaped = rows.reshape(nrows // group, group, 1024) if group else rows
    results = []
    for prefiltered in (True, False):
        itp._GEOSPLINE_PREFILTERED = prefiltered
        try:
            got = itp._fftlog_then_geospline(fft, s, r, shaped, dev, sqrt=False, group=group)
        finally:
            itp._GEOSPLINE_PREFILTERED = True
        assert got is not None
        results.append(got.permute(0, 2, 1).reshape(nrows, nq) if group else got)
    a, b = results
    assert bool(torch.isfinite(a).all()) and bool(torch.isfinite(b).all())
    scale = (fft(rows)[1].abs() * torch.as_tensor(s**1.5, device=dev)[None, :]).max(dim=1).values      # the row's own tilted scale
    tilted = ((a - b).abs() * torch.as_tensor(r**1.5, device=dev)[None, :]).max(dim=1).values / scale
    assert float(tilted.max()) < 2e-13, 'row %d' % int(tilted.argmax())


@pytest.mark.parametrize('shape,transposed', [((1024,), False), ((16, 64), True), ((600,), False), ((9, 70), True)])
def test_medium_batches_take_the_prefiltered_kernel(shape, transposed, monkeypatch):
    """sigma_r2_of_rows between 513 and 8192 rows: the prefiltered FFTLog + B-spline kernel (faster than the band operator inside the transform's kernel from
    ~500 rows on) -- same numbers as the band route to the agreement of two evaluations of an FFTLog, in both layouts; radii it refuses fall back."""
    torch, _lib, lib, dv, dev = _env()
    from cosmoprimo_amd import interpolator as itp
    rng = np.random.default_rng(sum(shape))
    r = np.geomspace(1., 120., 100)
    amp = torch.as_tensor(rng.uniform(0.5, 2., shape + (1,)), device=dev)

    def rows(k):
        return (amp * torch.as_tensor(1e4 * (k / 0.05)**-1.9 / (1. + (k / 0.3)**2), device=dev)).contiguous()

    def run(radii):
        for key in [key for key in itp._op_cache if key[0] == 'geospline']:
            del itp._op_cache[key]
        out = itp.sigma_r2_of_rows(radii, rows, device=dev, sqrt=True, radii_before_last_axis=transposed)
        return out, any(key[0] == 'geospline' and v.prefiltered for key, v in itp._op_cache.items())

    new, used = run(r)
    assert used
    monkeypatch.setattr(itp, '_GEOSPLINE_PREFILTERED_MIN_ROWS', 1 << 40)
    old, used = run(r)
    assert not used and new.shape == old.shape == ((shape[0], r.size, shape[1]) if transposed else shape + (r.size,))
    np.testing.assert_allclose(new.cpu().numpy(), old.cpu().numpy(), rtol=1e-12)
    monkeypatch.undo()
    near_the_end = np.array([0.012, 8.])      # within 32 knots of the first output point: the prefiltered plan is refused, the band operator answers
    got, used = run(near_the_end)
    assert not used and bool(torch.isfinite(got).all())


@pytest.mark.parametrize('kw', [dict(q=0.4), dict(q=-0.3, lowring=False), dict(q=0., lowring=False, xy=2.5)])
def test_prefiltered_geospline_other_transforms(kw):
    """The prefilter depends on the transform through the ratio of its output grid and the power law of its postfactor only: other tilts, no low-ringing
    condition, another product x y -- against scipy's natural spline of the package's own transform; radii unsorted, repeated, some outside the grid."""
    from scipy.interpolate import CubicSpline
    torch, _lib, lib, dv, dev = _env()
    import cosmoprimo_amd as cp
    from cosmoprimo_amd import interpolator as itp
    k = np.geomspace(1e-6, 1e2, 1024)
    fft = cp.TophatVariance(k, device=dev, **kw)
    s = fft.y[0]
    rng = np.random.default_rng(7)
    r = np.concatenate([rng.permutation(np.geomspace(s[40], s[400], 90)), [s[200], s[200], s[0] / 2., s[-1] * 3.]])
    rows = torch.as_tensor(rng.uniform(0.5, 2., (10, 1)) * 1e4 * (k / 0.05)**-1.9 / (1. + (k / 0.3)**2), device=dev).contiguous()
    got = itp._fftlog_then_geospline(fft, s, r, rows, dev)
    plan = [v for key, v in itp._op_cache.items() if key[0] == 'geospline'][-1]
    assert got is not None and plan.prefiltered
    got, var = got.cpu().numpy(), fft(rows)[1].cpu().numpy()
    inside = (r >= s[0]) & (r <= s[-1])
    assert np.isnan(got[:, ~inside]).all() and np.isfinite(got[:, inside]).all()
    tilt = 1.5 + kw['q']
    for i in range(rows.shape[0]):
        ref = CubicSpline(s, var[i], bc_type='natural')(r[inside])
        assert (np.abs(got[i, inside] - ref) * r[inside]**tilt).max() < 1e-13 * np.abs(var[i] * s**tilt).max(), 'row %d' % i


def test_geospline_plans_the_library_refuses():
    """Radii within 32 knots of either end of the grid, spans of more than 448 knots, grids that are not geometric, transforms of another size:
    no plan (the caller takes the band operator), never a wrong number."""
    torch, _lib, lib, dv, dev = _env()
    import cosmoprimo_amd as cp
    from cosmoprimo_amd import interpolator as itp
    k = np.geomspace(1e-7, 1e2, 1024)
    fft = cp.TophatVariance(k, device=dev)
    s = fft.y[0]
    rows = torch.as_tensor((k / 0.05)**-2. * 1e4, device=dev)[None, :].repeat(4, 1).contiguous()
    assert itp._GeoSpline(s, np.array([8.]), dev).handle is not None
    assert itp._GeoSpline(s, np.array([s[5] * 1.01, 8.]), dev).handle is None                 # 5 knots from the first one
    assert itp._GeoSpline(s, np.array([8., s[-20]]), dev).handle is None
    assert itp._GeoSpline(s, np.geomspace(s[100], s[900], 64), dev).handle is None            # 800 knots
    assert itp._GeoSpline(np.linspace(1., 2., 1024), np.array([1.5]), dev).handle is None
    assert itp._fftlog_then_geospline(fft, s, np.array([s[5] * 1.01, 8.]), rows, dev) is None
    other = cp.TophatVariance(np.geomspace(1e-5, 1e2, 512), device=dev)
    assert itp._fftlog_then_geospline(other, other.y[0], np.array([8.]), rows[:, :512].contiguous(), dev) is None
    with pytest.raises(ValueError):
        plan = itp._GeoSpline(s, np.array([8.]), dev)
        out = torch.empty((4, 1), dtype=torch.float64, device=dev)
        _lib.check(lib.cp_fftlog_geospline_execute(fft._get_plan(dev).handle, plan.handle, rows.data_ptr(), out.data_ptr(), 4, 3, 0, dv.stream_of(dev)))
    # the prefiltered plan: the same limits on the radii (the solve would be refused, the caller takes the band operator); a postfactor that is
    # no power law leaves the solve on the CU; such a plan runs its OWN transform and says so when handed another
    assert itp._GeoSpline(s, np.array([8.]), dev, fft=fft).prefiltered
    for radii in (np.array([s[5] * 1.01, 8.]), np.array([8., s[-20]]), np.geomspace(s[100], s[900], 64)):
        assert itp._GeoSpline(s, radii, dev, fft=fft).handle is None
    bent = cp.TophatVariance(k, device=dev)
    bent.padded_postfactor = bent.padded_postfactor * (1. + 1e-3 * np.sin(np.arange(2048) / 100.))
    plan = itp._GeoSpline(s, np.array([8.]), dev, fft=bent)
    assert plan.handle is not None and not plan.prefiltered
    plan = itp._GeoSpline(s, np.array([8.]), dev, fft=fft)
    out = torch.empty((4, 1), dtype=torch.float64, device=dev)
    with pytest.raises(ValueError, match='its own transform'):
        _lib.check(lib.cp_fftlog_geospline_execute(fft._get_plan(dev).handle, plan.handle, rows.data_ptr(), out.data_ptr(), 4, 0, 0, dv.stream_of(dev)))
    with pytest.raises(ValueError):
        _lib.check(lib.cp_fftlog_geospline_execute(None, itp._GeoSpline(s, np.array([8.]), dev).handle, rows.data_ptr(), out.data_ptr(), 4, 0, 0, dv.stream_of(dev)))


def test_bao_elementwise_passes():
    torch, _lib, lib, dv, dev = _env()
    rng = np.random.default_rng(5)
    st = dv.stream_of(dev)
    T = lambda a: torch.as_tensor(a, device=dev).contiguous()      # noqa: E731
    # wallish2018: pknow = a + b; out = pk / ((pk / pknow - 1) tophat + 1)
    nrows, n = 37, 1024
    pk, top = T(rng.uniform(0.5, 2., (nrows, n))), T(rng.uniform(0., 1., n))
    a = pk * T(rng.uniform(0.3, 0.6, (nrows, n)))
    b = pk * T(rng.uniform(0.97, 1.03, (nrows, n))) - a            # a + b = the smooth spectrum, within a few per cent of pk (the wiggles)
    out = torch.empty_like(pk)
    _lib.check(lib.cp_wallish_finish(pk.data_ptr(), a.data_ptr(), b.data_ptr(), top.data_ptr(), out.data_ptr(), nrows, n, 0, st))
    ref = pk / ((pk / (a + b) - 1.) * top + 1.)
    assert float(((out - ref) / ref).abs().max()) < 1e-14
    _lib.check(lib.cp_wallish_finish(pk.data_ptr(), a.data_ptr(), None, top.data_ptr(), out.data_ptr(), nrows, n, 0, st))
    assert float(((out - pk / ((pk / a - 1.) * top + 1.)) / ref).abs().max()) < 1e-13
    # brieden2022: ratio, padded knots, finish
    nb, m, nk, first = 23, 341, 1024, 300
    rows, now = T(rng.uniform(0.5, 2., (nb, m))), T(rng.uniform(0.5, 2., (nb, m)))
    g0, corr, rfid, rnow = T(rng.uniform(0.5, 2., nb)), T(rng.uniform(0.9, 1.1, m)), T(rng.uniform(0.9, 1.1, m)), T(rng.uniform(0.9, 1.1, m))
    pknow, ratio = torch.empty_like(rows), torch.empty_like(rows)
    _lib.check(lib.cp_brieden_ratio(rows.data_ptr(), now.data_ptr(), g0.data_ptr(), corr.data_ptr(), rfid.data_ptr(), pknow.data_ptr(), ratio.data_ptr(), nb, m, 0, st))
    ref_pknow = now * g0[:, None] * corr
    assert torch.equal(pknow, ref_pknow) and float((ratio / (rows / ref_pknow / rfid) - 1.).abs().max()) < 1e-15
    k_fid, rescale = np.geomspace(1e-3, 1., m), rng.uniform(0.9, 1.1, nb)
    env = T(rng.uniform(0.9, 1.1, (nb, m)))
    xk, yk = torch.empty((m + 4, nb), dtype=torch.float64, device=dev), torch.empty((m + 4, nb), dtype=torch.float64, device=dev)
    _lib.check(lib.cp_brieden_knots(env.data_ptr(), pknow.data_ptr(), rnow.data_ptr(), T(k_fid).data_ptr(), T(rescale).data_ptr(), 1e-7, 1e2, xk.data_ptr(),
                                    yk.data_ptr(), nb, m, 0, st))
    from cosmoprimo_amd.interpolator import _pad_log
    cols = (env * pknow * rnow).cpu().numpy()
    for c in (0, 11, nb - 1):      # _pad_log of the package's host path (pinned against the reference's in tests/test_oracle_interp.py) on the column's own knots
        lk, lp = _pad_log(k_fid / rescale[c], cols[c], extrap_kmin=1e-7, extrap_kmax=1e2)
        np.testing.assert_allclose(xk[:, c].cpu().numpy(), lk, rtol=1e-14, atol=1e-14)
        np.testing.assert_allclose(yk[:, c].cpu().numpy(), lp, rtol=1e-13, atol=1e-13)
    pkrows, res = T(rng.uniform(0.5, 2., (nb, nk))), T(rng.uniform(-1., 1., (m, nb)))
    out = torch.empty_like(pkrows)
    _lib.check(lib.cp_brieden_finish(pkrows.data_ptr(), res.data_ptr(), out.data_ptr(), nb, nk, first, m, 0, st))
    ref = pkrows.clone()
    ref[:, first:first + m] = (10.**res).T
    assert float((out / ref - 1.).abs().max()) < 1e-15
    assert lib.cp_brieden_finish(pkrows.data_ptr(), res.data_ptr(), out.data_ptr(), nb, nk, nk - 10, m, 0, st) != 0      # the range must fit the row


@pytest.mark.parametrize('path', ['valu', 'mfma'])
def test_spline_apply_grouped(path):
    """cp_spline_apply_grouped: rows taken in groups, the group index the fastest axis of the result (sigma_rz's (nz, nr) -> (nr, nz) inside the store)."""
    torch, _lib, lib, dv, dev = _env()
    from cosmoprimo_amd.spline import LinearOperator
    rng = np.random.default_rng(2)
    op = LinearOperator.spline(np.linspace(0., 1., 300), np.sort(rng.uniform(0., 1., 77)), bc='natural', device=dev)
    y = torch.as_tensor(rng.normal(size=(5, 12, 300)), device=dev)
    plain = op(y, path=path)                                   # (5, 12, 77)
    grouped = op(y, path=path, last_axis_first=True)           # (5, 77, 12)
    assert tuple(grouped.shape) == (5, 77, 12) and torch.equal(grouped, plain.transpose(-1, -2).contiguous())
    out = torch.empty((60 * 77,), dtype=torch.float64, device=dev)
    assert lib.cp_spline_apply_grouped(op._handle, y.data_ptr(), out.data_ptr(), 60, 7, 0, 1., dv.stream_of(dev)) != 0      # 60 rows are not groups of 7


@pytest.mark.parametrize('n', [1024, 2048])
def test_wallish_dd_box(n):
    """cp_wallish_dd_box (tridiagonal solve in LDS + the two arg-max searches) against scipy's clamped CubicSpline(x, nu=2) and numpy's argmax
    rule (reference bao_filter.py:377-394), on DST-like sequences and on rows with ties, NaN and a maximum at the edge of the search range."""
    import ctypes
    import torch
    from scipy import interpolate
    from cosmoprimo_amd import _lib, _device as dv
    rng = np.random.default_rng(n)
    nseq = 37
    x = 1. + np.arange(n)
    y = rng.normal(size=(nseq, n)) / x**1.5 + 3e-3 * np.exp(-0.5 * ((x - 0.35 * n) / 12.)**2) * rng.uniform(0.5, 2., size=(nseq, 1))
    y[3, 40:60] = 0.        # flat stretch: ties
    y[5, 100] = np.nan
    y[7] = 0.
    dev = torch.device('cuda', 0)
    ty = torch.as_tensor(y, device=dev)
    dd = torch.empty_like(ty)
    box = torch.empty((nseq, 2), dtype=torch.int32, device=dev)
    mf, ms, off = 20, 5, (-10, 20)
    lib = _lib.load()
    _lib.check(lib.cp_wallish_dd_box(ty.data_ptr(), nseq, n, mf, ms, off[0], off[1], box.data_ptr(), dd.data_ptr(), None, 0, dv.stream_of(dev)))
    box2 = torch.empty_like(box)
    filled = ty.clone()      # the removal of the box in the same kernel against cp_gap_spline on the same box
    _lib.check(lib.cp_wallish_dd_box(ty.data_ptr(), nseq, n, mf, ms, off[0], off[1], box2.data_ptr(), None, filled.data_ptr(), 0, dv.stream_of(dev)))
    separately = torch.empty_like(ty)
    _lib.check(lib.cp_gap_spline(ty.data_ptr(), box.data_ptr(), separately.data_ptr(), nseq, n, 0, dv.stream_of(dev)))
    keep = [i for i in range(nseq) if i != 5]
    np.testing.assert_allclose(filled.cpu().numpy()[keep], separately.cpu().numpy()[keep], rtol=1e-9, atol=1e-13 * np.abs(y[keep]).max())
    assert not torch.equal(filled, ty)
    got, gbox = dd.cpu().numpy(), box.cpu().numpy()
    assert np.array_equal(gbox, box2.cpu().numpy())
    for i in range(nseq):
        if i == 5:
            continue
        ref = interpolate.CubicSpline(x, y[i], bc_type='clamped')(x, nu=2)
        np.testing.assert_allclose(got[i], ref, rtol=1e-9, atol=1e-13 * np.abs(ref).max() + 1e-300)
        first = got[i][mf:-mf].argmax() + mf
        second = first + ms + got[i][first + ms:-mf].argmax()
        assert tuple(gbox[i]) == (first + off[0], second + off[1]), i
    assert np.isnan(got[5]).any()
    first = np.argmax(got[5][mf:-mf]) + mf      # numpy: NaN is the maximum
    assert gbox[5][0] == first + off[0]


def test_spliced_clamped_spline():
    """cp_splice_* (the clamped spline of wallish2018 through knots spliced from two arrays: tridiagonal solve in LDS, evaluation, damping) against
    scipy's CubicSpline(bc_type='clamped') on the gathered knots (reference bao_filter.py:415-431), for the filter's own grids and for knots
    without a uniform stretch; a row holding NaN stays alone."""
    import torch
    from scipy import interpolate
    from cosmoprimo_amd.spline import SplicedClampedSpline
    rng = np.random.default_rng(2)
    dev = torch.device('cuda', 0)
    k = np.geomspace(1e-7, 1e2, 1024)
    klin = np.linspace(1e-7, 2., 4096)
    mask = (klin > 1e-2) & (klin < 1.5)
    left, right = k < 5e-4, k > 2.
    knots = np.concatenate([k[left], klin[mask], k[right]])
    pieces = [(0, 0, int(left.sum())), (1, int(np.flatnonzero(mask)[0]), int(mask.sum())), (0, int(np.flatnonzero(right)[0]), int(right.sum()))]
    nrows = 11
    amp = rng.uniform(0.5, 2., size=(nrows, 1))
    shape = lambda x: x / (1. + (x / 0.02)**2.6)
    pk = amp * shape(k) * (1. + 0.05 * np.sin(k[None, :] / 0.01) * np.exp(-(k / 0.3)**2))
    lin = amp * shape(klin) * (1. + 1e-3 * rng.normal(size=(nrows, klin.size)))
    tophat = np.ones_like(k)
    tophat[k > 1.] *= np.exp(-20.**2 * (k[k > 1.] - 1.)**2)
    op = SplicedClampedSpline(knots, pieces, k, device=dev)
    assert op.scheme == 1      # 3 051 of the 3 666 knots on a uniform grid, every spline query inside: the recursions of cp_splice_uniform.h
    refs = [interpolate.CubicSpline(knots, np.concatenate([pk[i][left], lin[i][mask], pk[i][right]]), bc_type='clamped', extrapolate=False)(k) for i in range(nrows)]
    for scheme in (1, 0):      # ... and the elimination in LDS on the same plan
        op.scheme = scheme
        assert op.scheme == scheme
        got = op(torch.as_tensor(pk, device=dev), torch.as_tensor(lin, device=dev)).cpu().numpy()
        damped = op(torch.as_tensor(pk, device=dev), torch.as_tensor(lin, device=dev), tophat=torch.as_tensor(tophat, device=dev)).cpu().numpy()
        for i in range(nrows):
            np.testing.assert_allclose(got[i], refs[i], rtol=1e-11)
            np.testing.assert_allclose(damped[i], pk[i] / ((pk[i] / refs[i] - 1.) * tophat + 1.), rtol=1e-11)
        bad = pk.copy()
        bad[4, 400] = np.nan      # (one of the knots in front of the uniform stretch)
        gotb = op(torch.as_tensor(bad, device=dev), torch.as_tensor(lin, device=dev)).cpu().numpy()
        assert np.isnan(gotb[4]).any() and np.array_equal(np.delete(gotb, 4, axis=0), np.delete(got, 4, axis=0))
    # smooth spectra (no noise on the linear grid: second derivatives eight orders of magnitude apart along a row), more rows than one launch's waves
    op.scheme = 1
    many = 4099
    amp = rng.uniform(0.5, 2., size=(many, 1))
    tilt = rng.uniform(-0.1, 0.1, size=(many, 1))
    pk, lin = amp * shape(k) * k**tilt, amp * shape(klin) * klin**tilt
    got = op(torch.as_tensor(pk, device=dev), torch.as_tensor(lin, device=dev), tophat=torch.as_tensor(tophat, device=dev)).cpu().numpy()
    for i in list(range(0, many, 173)) + [many - 1]:
        ref = interpolate.CubicSpline(knots, np.concatenate([pk[i][left], lin[i][mask], pk[i][right]]), bc_type='clamped', extrapolate=False)(k)
        np.testing.assert_allclose(got[i], pk[i] / ((pk[i] / ref - 1.) * tophat + 1.), rtol=1e-9)
    # other grids: fewer queries, a stretch that ends the knots (clamped end next to it), knots of ONE array
    for nk, kmax_lin, nlin in ((512, 2., 3600), (1024, 1.2, 3000), (640, 3., 3500)):
        k2 = np.geomspace(1e-6, 50., nk)
        klin2 = np.linspace(1e-6, kmax_lin, nlin)
        m2 = (klin2 > 2e-2) & (klin2 < 0.9 * kmax_lin)
        l2, r2 = k2 < 1e-3, k2 > kmax_lin
        knots2 = np.concatenate([k2[l2], klin2[m2], k2[r2]])
        pieces2 = [(0, 0, int(l2.sum())), (1, int(np.flatnonzero(m2)[0]), int(m2.sum())), (0, int(np.flatnonzero(r2)[0]), int(r2.sum()))]
        op2 = SplicedClampedSpline(knots2, pieces2, k2, device=dev)
        assert op2.scheme == 1, (nk, kmax_lin, nlin)
        amp = rng.uniform(0.5, 2., size=(9, 1))
        pk2, lin2 = amp * shape(k2), amp * shape(klin2) * (1. + 1e-4 * rng.normal(size=(9, nlin)))
        got = op2(torch.as_tensor(pk2, device=dev), torch.as_tensor(lin2, device=dev)).cpu().numpy()
        for i in range(9):
            ref = interpolate.CubicSpline(knots2, np.concatenate([pk2[i][l2], lin2[i][m2], pk2[i][r2]]), bc_type='clamped', extrapolate=False)(k2)
            np.testing.assert_allclose(got[i], ref, rtol=1e-10)
    # a plan whose queries need the spline far from the uniform stretch keeps the elimination
    xk = np.concatenate([np.geomspace(1e-4, 9e-3, 200), np.linspace(1e-2, 1., 1500)])
    opq = SplicedClampedSpline(xk, [(0, 0, xk.size)], np.geomspace(2e-4, 0.9, 300), device=dev)
    assert opq.scheme == 0
    with pytest.raises(Exception):
        opq.scheme = 1
    # irregular knots from one array, queries inside and outside
    x = np.sort(rng.uniform(0., 10., 700))
    xq = np.concatenate([[-1.], rng.uniform(x[0], x[-1], 300), [x[0], x[-1], 11.]])
    y = rng.normal(size=(5, 700))
    op = SplicedClampedSpline(x, [(0, 0, 700)], xq, device=dev)
    got = op(torch.as_tensor(y, device=dev)).cpu().numpy()
    for i in range(5):
        ref = interpolate.CubicSpline(x, y[i], bc_type='clamped', extrapolate=False)(xq)
        assert np.isnan(got[i][0]) and np.isnan(got[i][-1])
        np.testing.assert_allclose(got[i][1:-1], ref[1:-1], rtol=1e-9, atol=1e-11)


@pytest.mark.parametrize('bc', ['natural', 'clamped', 'not-a-knot'])
def test_spline_rows(bc):
    """cp_spline_rows_* (the spline's tridiagonal system solved per row in LDS, only the knots the queries can see) against scipy's CubicSpline and
    against the operator route, for the three rows-per-wave layouts: a short window of a long geometric grid (the FFTLog output -> radii step of
    sigma_r), a medium one, and all knots; root, scale and the grouped store; queries outside the knots; a row holding NaN stays alone."""
    import torch
    from scipy import interpolate
    from cosmoprimo_amd.spline import SplineRows, LinearOperator
    rng = np.random.default_rng(5)
    dev = torch.device('cuda', 0)
    outside = lambda q: np.concatenate([[1e-3], q, [2e7]])
    cases = [(1024, np.geomspace(1., 100., 256), 4), (1024, np.geomspace(0.1, 3e3, 300), 2), (1024, outside(np.geomspace(1e-2, 1e7, 500)), 2),
             (2000, outside(np.geomspace(1e-2, 1e7, 700)), 1)]
    for n, xq, rows_per_wave in cases:
        x = np.geomspace(1e-2, 1e7, n)
        op = SplineRows(x, xq, bc=bc, device=dev)
        first, nknots, rpw, halo = op.window
        assert rpw == rows_per_wave, (op.window, rows_per_wave)
        y = rng.uniform(0.5, 1.5, size=(3, 7, n)) * (1. + x / 50.)**-1.5
        ty = torch.as_tensor(y, device=dev)
        got = op(ty).cpu().numpy()
        ref = interpolate.CubicSpline(x, y, axis=-1, bc_type=bc, extrapolate=False)(xq)
        inside = (xq >= x[0]) & (xq <= x[-1])
        assert np.isnan(got[..., ~inside]).all()
        np.testing.assert_allclose(got[..., inside], ref[..., inside], rtol=1e-11)
        banded = LinearOperator.spline(x, xq, bc=bc, device=dev)
        np.testing.assert_allclose(got[..., inside], banded(ty).cpu().numpy()[..., inside], rtol=1e-11)
        rooted = op(ty, sqrt=True, scale=0.25, last_axis_first=True).cpu().numpy()      # (3, nq, 7)
        np.testing.assert_allclose(rooted[:, inside], np.sqrt(0.25 * ref).transpose(0, 2, 1)[:, inside], rtol=1e-11)
        bad = y.copy()
        bad[1, 3, first + nknots // 2] = np.nan
        gotb = op(torch.as_tensor(bad, device=dev)).cpu().numpy()
        mask = np.ones(got.shape[:2], dtype=bool)
        mask[1, 3] = False
        assert np.array_equal(gotb[mask], got[mask], equal_nan=True)
    # the second derivatives at the knots (the spline's own representation)
    x = np.sort(np.concatenate([[0.], rng.uniform(0., 9., 502), [9.]]))
    y = rng.normal(size=(13, x.size))
    op = SplineRows(x, x, bc=bc, device=dev)
    got = op.second_derivatives(torch.as_tensor(y, device=dev)).cpu().numpy()
    ref = interpolate.CubicSpline(x, y, axis=-1, bc_type=bc)(x, nu=2)
    np.testing.assert_allclose(got, ref, rtol=1e-8, atol=1e-9 * np.abs(ref).max())
    # ... written beside the knot values, (y_j, M_j): the same numbers
    pairs = op.second_derivatives(torch.as_tensor(y, device=dev), pairs=True).cpu().numpy()
    assert pairs.shape == y.shape + (2,) and np.array_equal(pairs[..., 0], y) and np.array_equal(pairs[..., 1], got)


@pytest.mark.parametrize('shape', [(3, 30, 504, 64, 1024), (2, 13, 211, 37, 1000), (1, 5, 64, 3, 70), (5, 32, 100, 64, 256)])
@pytest.mark.parametrize('post', [None, 'exp10'])
def test_tables_rows_direct(shape, post):
    """cp_tables_rows_direct (the k splines evaluated from the tables' second derivatives, the z contraction on the matrix cores) against the two
    operator launches and against scipy's two passes (jax.py:241-271): odd shapes, wavenumbers outside the knots, a table holding NaN."""
    torch, _lib, lib, dv, dev = _env()
    from scipy.interpolate import CubicSpline
    from cosmoprimo_amd.spline import LinearOperator, SplineRows, dense_operator
    nb, nzin, n, nzq, nq = shape
    rng = np.random.default_rng(nb + n)
    x = np.linspace(0., 10., n) + rng.uniform(-0.3, 0.3, n) * 10. / n
    x[0], x[-1] = 0., 10.
    xq = np.sort(np.concatenate([[0., 10.], rng.uniform(-0.5, 10.5, nq - 2)]))               # the ends of the knots, and some queries outside them: NaN
    zk, zq = np.linspace(0., 3., nzin), np.linspace(0., 3., nzq)
    kplan = SplineRows(x, xq, bc='not-a-knot', device=dev)
    opx = LinearOperator.spline(x, xq, bc='not-a-knot', extrapolate=False, device=dev)
    opz = LinearOperator.dense(dense_operator(zk, zq, bc='not-a-knot', extrapolate=True), device=dev)
    smooth = 1. + 0.3 * np.sin(x)[None, None, :] * np.cos(zk)[None, :, None] * rng.uniform(0.5, 1., (nb, 1, 1))
    t = torch.as_tensor(smooth + 1e-3 * rng.standard_normal((nb, nzin, n)), device=dev)
    m = SplineRows(x, x[[0, -1]], bc='not-a-knot', device=dev).second_derivatives(t)
    out = torch.full((nb, nzq, nq), -7., dtype=torch.float64, device=dev)
    code = {None: 0, 'exp10': 2}[post]
    _lib.check(lib.cp_tables_rows_direct(kplan._handle, opz._handle, t.data_ptr(), m.data_ptr(), out.data_ptr(), nb, code, 1., dv.stream_of(dev)))
    got, ref = out.cpu().numpy(), opz.mid(opx(t), post=post).cpu().numpy()
    # tables and second derivatives as (y, M) pairs in one array (d_m null): the same operations on the same numbers
    ym = SplineRows(x, x[[0, -1]], bc='not-a-knot', device=dev).second_derivatives(t, pairs=True)
    out2 = torch.full((nb, nzq, nq), -7., dtype=torch.float64, device=dev)
    _lib.check(lib.cp_tables_rows_direct(kplan._handle, opz._handle, ym.data_ptr(), None, out2.data_ptr(), nb, code, 1., dv.stream_of(dev)))
    assert np.array_equal(out2.cpu().numpy(), got, equal_nan=True)
    outside = (xq < x[0]) | (xq > x[-1])
    assert np.isnan(got[..., outside]).all() and np.isfinite(got[..., ~outside]).all()
    np.testing.assert_allclose(got[..., ~outside], ref[..., ~outside], rtol=1e-11, atol=1e-12)
    if nzin >= 4:
        two = CubicSpline(zk, CubicSpline(x, t[0].cpu().numpy(), axis=1)(xq[~outside]), axis=0)(zq)
        np.testing.assert_allclose(got[0][:, ~outside], two if post is None else 10.**two, rtol=1e-10, atol=1e-12)
    if nb > 1:
        t2 = t.clone()
        t2[1, nzin // 2, n // 3] = float('nan')
        m2 = SplineRows(x, x[[0, -1]], bc='not-a-knot', device=dev).second_derivatives(t2)
        _lib.check(lib.cp_tables_rows_direct(kplan._handle, opz._handle, t2.data_ptr(), m2.data_ptr(), out.data_ptr(), nb, code, 1., dv.stream_of(dev)))
        again = out.cpu().numpy()
        assert np.isnan(again[1]).any() and np.array_equal(np.delete(again, 1, axis=0), np.delete(got, 1, axis=0), equal_nan=True)


def test_round4_entry_points_take_empty_batches():
    """Zero rows / cosmologies / pairs through every entry point added in round 4: CP_OK, nothing launched, outputs of the right (empty) shape."""
    torch, _lib, lib, dv, dev = _env()
    import cosmoprimo_amd as cp
    from cosmoprimo_amd import interpolator as itp
    from cosmoprimo_amd.dst import DST
    from cosmoprimo_amd.fftlog import NumpyFFTEngine
    k = np.geomspace(1e-7, 1e2, 1024)
    fft = cp.TophatVariance(k, device=dev)
    empty = torch.empty((0, 1024), dtype=torch.float64, device=dev)
    out = itp._fftlog_then_geospline(fft, fft.y[0], np.array([8., 20.]), empty, dev)
    assert out is not None and tuple(out.shape) == (0, 2)
    grouped = itp._fftlog_then_geospline(fft, fft.y[0], np.array([8., 20.]), empty.reshape(0, 4, 1024), dev, group=4)
    assert grouped is not None and tuple(grouped.shape) == (0, 2, 4)
    assert NumpyFFTEngine(64).forward(np.empty((0, 64))).shape == (0, 33)
    assert NumpyFFTEngine(64).backward(np.empty((3, 0, 33), dtype='c16')).shape == (3, 0, 64)
    for name, args in (('cp_derived_parameters', (0, None, None, 0, None)),
                       ('cp_bilinear_pairs', (None, None, None, None, 0, 5, 4, 3, 0, None)),
                       ('cp_variants_scalars', (0, None, 0, None, None, 0, None))):
        assert getattr(lib, name)(*args) == _lib.CP_OK, name
    none = torch.empty(0, dtype=torch.float64, device=dev)
    bg = dict(h=none, Omega_cdm=none, Omega_b=none)
    rs, amp, spectra, kk = itp.sigma8_normalise('eisenstein_hu', bg, dict(n_s=none), 0.8, dev)
    assert tuple(rs.shape) == (0,) and tuple(amp.shape) == (0,) and tuple(spectra.shape) == (0, 1024) and kk.shape == (1024,)
    coefficients = DST(4096, kx=np.linspace(1e-4, 5., 4096), device=dev).forward_analytic('eisenstein_hu', bg, dict(n_s=none))
    assert tuple(coefficients.shape) == (0, 4096)
