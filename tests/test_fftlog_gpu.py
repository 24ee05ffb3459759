"""GPU parity tests of the fused HIP FFTLog kernel (through the C ABI) against the oracle and the reference's golden
vectors.  Tolerances: (i) tilted-space norm-wise max|d(g y^q)| / max|g y^q| <= 1e-13; (ii) pointwise
relative <= 1e-10 for s in [1e-2, 2e2] Mpc/h; outside that range values are 16 decades below peak and are not gated."""
import numpy as np
import pytest

from conftest import tilted_err
from oracle import fftlog as ofl

pytestmark = pytest.mark.gpu

TOL_NORM = 1e-13
TOL_POINT = 1e-10


@pytest.fixture(scope='module')
def cp():
    import torch
    assert torch.cuda.is_available(), 'GPU tests need a ROCm device'
    import cosmoprimo_amd
    return cosmoprimo_amd


def pointwise(g, ref, s, lo=1e-2, hi=2e2):
    m = (s > lo) & (s < hi)
    return np.abs(g[..., m] / ref[..., m] - 1.).max()


@pytest.mark.parametrize('n', [1024, 2048])
def test_golden_transforms(cp, golden, n):
    """G3: default-cosmology EH P(k) (configs 1 and 2 grids), every extrap mode, keep_padding, multi-ell, tophat, xi->P."""
    g = golden('fftlog_transforms')
    pkd = golden('pk_eh_default')
    k, pk = pkd['k%d' % n], pkd['pk%d' % n]
    f = cp.PowerToCorrelation(k, ell=0, lowring=True)
    for name, extrap in [('zero', 0), ('edge', 'edge'), ('log', 'log'), ('mixed', ('log', 0.)), ('const', (1.5, 'edge'))]:
        s, xi = f(pk, extrap=extrap)
        ref = g['n%d_p2c_l0_%s' % (n, name)]
        assert xi.shape == (n,) and xi.dtype == np.float64
        np.testing.assert_allclose(s, g['n%d_p2c_l0_s' % n], rtol=1e-14)
        # 'edge' / constant padding put the largest tilted samples in the padding: conditioning ~1e2 worse (see oracle test)
        tol = TOL_NORM if name in ('zero', 'log', 'mixed') else 1e-11
        assert tilted_err(xi, ref, s, 1.5) < tol, name
        if name in ('zero', 'log', 'mixed'):
            assert pointwise(xi, ref, s) < TOL_POINT, name
    y, keep = f(pk, extrap='log', keep_padding=True)
    assert keep.shape == (2 * n,) and y.shape == (2 * n,)
    assert tilted_err(keep, g['n%d_p2c_l0_keep' % n], y, 1.5) < TOL_NORM
    fm = cp.PowerToCorrelation(k, ell=[0, 2, 4], lowring=True)
    s3, xi3 = fm(pk)   # 1-D input broadcast against 3 kernels
    assert xi3.shape == (3, n) and s3.shape == (3, n)
    for i in range(3):
        assert tilted_err(xi3[i], g['n%d_p2c_l024' % n][i], s3[i], 1.5) < TOL_NORM
    r, var = cp.TophatVariance(k)(pk)
    assert tilted_err(var, g['n%d_tophat' % n], r, 1.5) < TOL_NORM
    assert pointwise(var, g['n%d_tophat' % n], r, 1e-1, 1e2) < TOL_POINT
    k2, pk2 = cp.CorrelationToPower(g['n%d_p2c_l0_s' % n], ell=0, lowring=True)(g['n%d_p2c_l0_zero' % n])
    assert tilted_err(pk2, g['n%d_c2p_l0' % n], k2, 1.5) < 1e-12


def test_config2_golden_rows(cp, golden):
    from oracle.workloads import config2_rows
    g = golden('fftlog_transforms')
    pkd = golden('pk_eh_default')
    k, pk = pkd['k2048'], pkd['pk2048']
    f = cp.PowerToCorrelation(k, ell=0)
    rows = np.concatenate([config2_rows(k, pk, i, i + 1) for i in g['config2_idx']])
    s, xi = f(rows)
    for a, b in zip(xi, g['config2_xi']):
        assert tilted_err(a, b, s, 1.5) < TOL_NORM
        assert pointwise(a, b, s) < TOL_POINT


@pytest.mark.parametrize('engine', ['numpy', 'fftw', 'mi355x'])
def test_hankel60_against_the_reference_numbers(cp, golden, engine):
    """Order-0 Hankel transform of the 60-sample (1 + x^2)^-3/2 the reference transformed for goldens (`hankel60_*`, oracle/gen_golden.py): the
    forward result, its closed form exp(-y), and the inverse transform of that result, for every engine name (all three select the fused
    kernel).  Tolerances: 1e-11 against the reference's forward numbers, 1e-10 against its inverse."""
    g = golden('fftlog_transforms')
    forward = cp.HankelTransform(g['hankel60_x'], nu=0, q=1, lowring=True, engine=engine)
    y, got = forward(g['hankel60_f'], extrap='log')
    np.testing.assert_allclose(y, g['hankel60_y'], rtol=1e-14)
    np.testing.assert_allclose(got, g['hankel60_g'], rtol=1e-11, atol=1e-14)
    assert np.abs(got - np.exp(-y)).max() < 2e-8              # the closed form, to the accuracy 60 samples give
    forward.inv()                                              # in place: the same object now maps g back to f
    x_back, f_back = forward(got, extrap='log')
    np.testing.assert_allclose(x_back, g['hankel60_inv_x'], rtol=1e-14)
    np.testing.assert_allclose(f_back, g['hankel60_inv_f'], rtol=1e-10, atol=1e-13)
    assert np.abs(f_back - g['hankel60_f']).max() < 1e-7


def test_hankel_of_a_stack_of_scaled_rows(cp):
    """The transform is linear and row-wise: exp(-y) on another 60-sample grid gives (1 + x^2)^-3/2 (1e-10), and a stack of scaled copies of the
    row gives the scaled results on ONE output grid."""
    y = np.logspace(-4, 2, num=60, endpoint=False)
    transform = cp.HankelTransform(y, nu=0, q=1, lowring=True)
    x, single = transform(np.exp(-y), extrap='log')
    closed_form = (1. + x**2)**-1.5
    assert np.abs(single - closed_form).max() < 2e-10
    factors = np.array([1., 2., 3.])
    x3, stack = transform(factors[:, None] * np.exp(-y)[None, :], extrap='log')
    assert x3.shape == (60,) and stack.shape == (3, 60)
    np.testing.assert_array_equal(x3, x)
    np.testing.assert_allclose(stack, factors[:, None] * single[None, :], rtol=1e-13, atol=1e-16)


def test_multipoles_there_and_back(cp, golden):
    """P -> xi_ell -> P for ell = 0 ... 4 on the golden EH98 spectrum: the round trip returns P where the transform is well conditioned (1 % on
    1e-2 < k < 10), the five multipoles in one call equal the five single calls, without low-ringing s k = 1 holds exactly, and the complex
    convention carries the (-i)^ell phases (oracle)."""
    pkd = golden('pk_eh_default')
    k, pk = pkd['k1024'], pkd['pk1024']
    ells = (0, 1, 2, 3, 4)
    singles = []
    for ell in ells:
        s, xi = cp.PowerToCorrelation(k, ell=ell, lowring=True, complex=False)(pk)
        assert xi.shape == (1024,) and np.abs(xi).max() > 0.
        singles.append(xi)
        k_back, pk_back = cp.CorrelationToPower(s, ell=ell, lowring=True, complex=False)(xi)
        window = (k_back > 1e-2) & (k_back < 10.)
        expected = np.exp(np.interp(np.log(k_back[window]), np.log(k), np.log(pk)))
        assert np.abs(pk_back[window] / expected - 1.).max() < 1e-2, ell
    _, together = cp.PowerToCorrelation(k, ell=list(ells), lowring=True, q=0, complex=False)(pk)
    np.testing.assert_allclose(together, np.array(singles), rtol=1e-5, atol=1e-8)      # (another tilt q: equal to the accuracy of the transform, not to rounding)
    s_plain, _ = cp.PowerToCorrelation(k, ell=0, lowring=False)(pk)
    np.testing.assert_allclose(s_plain[::-1] * k, 1., rtol=1e-13)
    sc, xic = cp.PowerToCorrelation(k, ell=list(ells), complex=True)(pk)
    assert xic.dtype == np.complex128
    t = ofl.power_to_correlation(k, ell=list(ells))
    ref = ofl.apply(t, pk) * ((-1.) ** (np.array(ells) // 2))[:, None] * ((-1j) ** np.array(ells))[:, None]
    for i in range(5):
        assert np.abs((xic[i] - ref[i]) * sc[i]**1.5).max() / np.abs(ref[i] * sc[i]**1.5).max() < TOL_NORM


@pytest.mark.parametrize('n', [2, 3, 5, 8, 13, 30, 60, 100, 250, 256, 500, 512, 1000, 1024, 2048, 3000, 4096])
def test_sizes_and_modes_vs_oracle(cp, n):
    """Every kernel size / variant, odd batches (incomplete pair), multi-kernel, keep_padding, all extrap modes."""
    rng = np.random.default_rng(n)
    k = np.logspace(-3, 2, n)
    t = ofl.power_to_correlation(k, ell=[0, 2])
    f = cp.PowerToCorrelation(k, ell=[0, 2])
    fun = rng.uniform(0.95, 1.05, size=(5, 2, n)) * k**-1.2
    modes = [(0, False), (0, True), ('edge', False), ((1.5, 'edge'), False), ((0., 'edge'), True)]
    if n >= 3 and n <= 2048:
        modes += [('log', True), (('log', 0.3), False), ((0.2, 'log'), False)]
    with np.errstate(all='ignore'):
        for extrap, keep in modes:
            ref = ofl.apply(t, fun, extrap=extrap, keep_padding=keep)
            y, got = f(fun, extrap=extrap, keep_padding=keep)
            assert got.shape == ref.shape
            scale = np.abs(ofl.pad(fun, (t.in_left, t.in_right), extrap) * t.pre).max(axis=-1)   # each row's OWN tilted magnitude
            post = np.abs(t.post[:, t.out_left:t.out_left + t.n] if not keep else t.post)
            err = np.abs(got - ref) / post / scale[..., None]
            assert err.max() < 2e-14, (n, extrap, keep, err.max())


def test_shapes_dtypes_devices(cp):
    import torch
    n = 128
    k = np.logspace(-3, 2, n)
    pk = k**-1.5
    f1 = cp.PowerToCorrelation(k, ell=0)
    f3 = cp.PowerToCorrelation(k, ell=[0, 2, 4])
    t1 = ofl.power_to_correlation(k, ell=0)
    s, xi = f1(pk)
    assert s.shape == (n,) and xi.shape == (n,)
    s, xi = f1(np.tile(pk, (7, 3, 1)))
    assert xi.shape == (7, 3, n)
    assert np.allclose(xi, ofl.apply(t1, pk)[0], rtol=1e-12)
    s, xi = f3(np.tile(pk, (7, 3, 1)))
    assert s.shape == (3, n) and xi.shape == (7, 3, n)
    s, xi = f3(np.tile(pk, (7, 1, 1)))       # (7, 1, n) broadcasts against 3 kernels
    assert xi.shape == (7, 3, n)
    s, xi = f1(pk.astype('f4'))              # f4 in -> f8 out
    assert xi.dtype == np.float64
    s, xi = f1(np.zeros((0, n)))             # empty batch
    assert xi.shape == (0, n)
    s, xi = f1(pk, keep_padding=True)
    assert s.shape == (2 * n,) and xi.shape == (2 * n,)
    # torch in -> torch out on the same device, no host round trip
    tpk = torch.as_tensor(np.tile(pk, (5, 1)), device='cuda')
    ts, txi = f1(tpk)
    assert isinstance(txi, torch.Tensor) and txi.is_cuda and txi.shape == (5, n) and txi.dtype == torch.float64
    assert np.allclose(txi.cpu().numpy(), ofl.apply(t1, pk)[0], rtol=1e-12)
    # a side stream is honoured
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        ts, txi2 = f1(tpk)
    st.synchronize()
    assert torch.equal(txi, txi2)
    with pytest.raises(ValueError):
        f1(np.ones(n + 1))
    with pytest.raises(ValueError):
        f1(pk, extrap='nope')


def test_full_size_batch_properties(cp, golden):
    """BASELINE config 2 at full size (100k x 2048): sampled rows against the oracle, linearity over the whole batch,
    and determinism (bitwise identical reruns)."""
    import torch
    pkd = golden('pk_eh_default')
    k, pk = pkd['k2048'], pkd['pk2048']
    nb = 100000
    rng = np.random.default_rng(0)
    amp, dn = rng.uniform(0.5, 2., nb), rng.uniform(-0.1, 0.1, nb)
    dev = torch.device('cuda')
    tk, tpk = torch.as_tensor(k, device=dev), torch.as_tensor(pk, device=dev)
    rows = torch.as_tensor(amp, device=dev)[:, None] * (tk[None, :] / 0.05) ** torch.as_tensor(dn, device=dev)[:, None] * tpk[None, :]
    f = cp.PowerToCorrelation(k, ell=0)
    t = ofl.power_to_correlation(k, ell=0)
    s, xi = f(rows)
    s, xi2 = f(rows)
    assert torch.equal(xi, xi2)
    idx = np.unique(np.concatenate([[0, 1, nb - 2, nb - 1], rng.integers(0, nb, 60)]))
    sub_in = rows[torch.as_tensor(idx, device=dev)].cpu().numpy()
    sub_out = xi[torch.as_tensor(idx, device=dev)].cpu().numpy()
    ref = ofl.apply(t, sub_in[:, None, :])[:, 0]
    sn = s.cpu().numpy()
    for a, b in zip(sub_out, ref):
        assert tilted_err(a, b, sn, 1.5) < TOL_NORM
        assert pointwise(a, b, sn) < TOL_POINT
    # linearity: sum over the batch of the outputs == transform of the summed input
    total_in = rows.sum(dim=0).cpu().numpy()
    total_out = xi.sum(dim=0).cpu().numpy()
    ref_total = ofl.apply(t, total_in)[0]
    assert tilted_err(total_out, ref_total, sn, 1.5) < 1e-12


def direct_execute(cp, f, rows, extrap=(0, 0., 0, 0.), keep=False):
    """cp_fftlog_execute called directly on device buffers (no facade in between): rows (nbatch, nker, n) -> same shape."""
    import torch
    from cosmoprimo_amd import _lib
    dev = torch.device('cuda', torch.cuda.current_device())
    tin = torch.as_tensor(np.ascontiguousarray(rows, dtype='f8'), device=dev)
    nbatch = tin.shape[0]
    tout = torch.full(tin.shape[:-1] + (f.padded_size if keep else f.size,), 7.5, dtype=torch.float64, device=dev)
    plan = f._get_plan(dev)
    _lib.check(_lib.load().cp_fftlog_execute(plan.handle, tin.data_ptr(), tout.data_ptr(), nbatch, extrap[0], extrap[1], extrap[2], extrap[3],
                                             int(keep), torch.cuda.current_stream(dev).cuda_stream))
    torch.cuda.synchronize()
    return tout.cpu().numpy()


@pytest.mark.parametrize('n', [1024, 2048, 256, 500, 6000])
def test_execute_window(cp, n):
    """cp_fftlog_execute_window: the columns inside the window are those of cp_fftlog_execute bit for bit; the default transform (zero padding,
    power-of-two rows) leaves the others as they were, every other case may write whole rows; odd batches, one and two kernels, NaN rows."""
    import torch
    from cosmoprimo_amd import _lib
    rng = np.random.default_rng(n)
    k = np.logspace(-3, 2, n)
    dev = torch.device('cuda', torch.cuda.current_device())
    for ell, codes in [(0, (0, 0., 0, 0.)), ([0, 2], (0, 0., 0, 0.)), (0, (1, 0., 1, 0.))]:
        f = cp.PowerToCorrelation(k, ell=ell)
        nker = 2 if isinstance(ell, list) else 1
        rows = rng.uniform(0.9, 1.1, size=(7, nker, n)) * k**-1.2
        rows[3, 0, 5] = np.nan
        with np.errstate(all='ignore'):
            full = direct_execute(cp, f, rows, codes)
        plan = f._get_plan(dev)
        tin = torch.as_tensor(rows, device=dev)
        for first, count in [(n // 6, n // 3), (0, 1), (n - 3, 3), (0, n), (n // 2 - 1, 2), (5, 0)]:
            tout = torch.full(tin.shape, 7.5, dtype=torch.float64, device=dev)
            _lib.check(_lib.load().cp_fftlog_execute_window(plan.handle, tin.data_ptr(), tout.data_ptr(), rows.shape[0], codes[0], codes[1], codes[2], codes[3], 0,
                                                            first, count, torch.cuda.current_stream(dev).cuda_stream))
            got = tout.cpu().numpy()
            np.testing.assert_array_equal(got[..., first:first + count], full[..., first:first + count])
            outside = np.ones(n, dtype=bool)
            outside[first:first + count] = False
            if codes[0] == 0 and n in (256, 1024, 2048):      # the windowed kernel
                assert (got[..., outside] == 7.5).all(), (n, ell, first, count)
            else:
                same = (got[..., outside] == full[..., outside]) | (np.isnan(got[..., outside]) & np.isnan(full[..., outside]))
                assert (same | (got[..., outside] == 7.5)).all()
        with pytest.raises(ValueError):
            _lib.check(_lib.load().cp_fftlog_execute_window(plan.handle, tin.data_ptr(), tout.data_ptr(), rows.shape[0], 0, 0., 0, 0., 0, n - 2, 3, None))
    # the facade: FFTlog.__call__(..., out_window=)
    f = cp.TophatVariance(k)
    rows = torch.as_tensor(rng.uniform(0.9, 1.1, size=(5, n)) * k**-1.2, device=dev)
    a, b = f(rows)[1], f(rows, out_window=(n // 4, n // 2))[1]
    assert torch.equal(a[:, n // 4:n // 4 + n // 2], b[:, n // 4:n // 4 + n // 2])


@pytest.mark.parametrize('n', [2048, 1024, 500, 100, 16, 6, 6000, 40000])
def test_rows_are_independent_through_the_c_abi(cp, golden, n):
    """The kernel itself keeps rows independent (reference: numpy transforms row by row, fftlog.py:538-544): rows spanning
    1e-12 ... 1e7 in one batch are each accurate relative to their OWN magnitude, and a NaN / Inf row gives a NaN row without
    touching its pair partner.  Direct cp_fftlog_execute calls; zero / edge / log padding, one and two kernels, odd batches."""
    rng = np.random.default_rng(n)
    k = np.logspace(-3, 2, n)
    amp = np.array([1., 1e-12, 3e7, 1e-3, 2., 1.5e-6, 1e7, 4e-9, 1.])   # 9 rows: pairs (0,1) (2,3) (4,5) (6,7) and a single
    for ell, codes, oext in [(0, (0, 0., 0, 0.), 0), ([0, 2], (0, 0., 0, 0.), 0), (0, (1, 0., 1, 0.), 'edge'), ([0, 2], (2, 0., 2, 0.), 'log')]:
        if oext == 'log' and (n < 3 or n > 4096):      # (the geometric continuation over thousands of padded samples overflows, in the reference as well)
            continue
        f = cp.PowerToCorrelation(k, ell=ell)
        t = ofl.power_to_correlation(k, ell=ell if isinstance(ell, list) else [ell])
        base = rng.uniform(0.9, 1.1, size=(amp.size, t.nker, n)) * k**-1.2
        rows = base * amp[:, None, None]
        with np.errstate(all='ignore'):
            ref = ofl.apply(t, rows, extrap=oext)
            got = direct_execute(cp, f, rows, codes)
            scale = np.abs(ofl.pad(rows, (t.in_left, t.in_right), oext) * t.pre).max(axis=-1)      # each row's own tilted magnitude
        post = np.abs(t.post[:, t.out_left:t.out_left + t.n])
        err = np.abs(got - ref) / post / scale[..., None]
        # rows within a factor 32 of their partner share rounding (no rescaling below that spread): 32 x 1e-15
        assert err.max() < 5e-14, (n, ell, oext, err.max(axis=(1, 2)))
        # non-finite rows: NaN out, partner untouched (bitwise the same as in the clean batch when no rescaling is involved)
        bad = rows.copy()
        bad[0, 0, n // 3] = np.nan
        bad[5, -1, 0] = np.inf
        bad[8, 0, n - 1] = -np.inf
        with np.errstate(all='ignore'):
            gotb = direct_execute(cp, f, bad, codes)
        for b, ker in [(0, 0), (5, t.nker - 1), (8, 0)]:
            assert np.isnan(gotb[b, ker]).all(), (n, ell, oext, b)
        clean = np.ones(got.shape[:2], dtype=bool)
        clean[0, 0] = clean[5, -1] = clean[8, 0] = False
        assert np.isfinite(gotb[clean]).all()
        errb = np.abs(gotb - ref)[clean] / np.broadcast_to(post, got.shape)[clean] / scale[clean][..., None]
        assert errb.max() < 5e-14, (n, ell, oext, errb.max())


def test_mixed_magnitude_batch_pointwise(cp, golden):
    """Rows spanning 1e-12 ... 1e7 (random order, so most pairs mix magnitudes): every row within 1e-10 pointwise of the oracle for
    s in [1e-2, 2e2] Mpc/h and within 1e-13 norm-wise in tilted space -- the same gates as for a homogeneous batch."""
    pkd = golden('pk_eh_default')
    k, pk = pkd['k2048'], pkd['pk2048']
    rng = np.random.default_rng(5)
    amp = 10.**rng.uniform(-12, 7, size=257)
    rows = amp[:, None] * pk * (k / 0.05)**rng.uniform(-0.1, 0.1, size=(amp.size, 1))
    t = ofl.power_to_correlation(k)
    ref = ofl.apply(t, rows[:, None, :])[:, 0]
    s = t.y[0]
    got = direct_execute(cp, cp.PowerToCorrelation(k), rows[:, None, :])[:, 0]
    assert max(tilted_err(got[i], ref[i], s, 1.5) for i in range(amp.size)) < TOL_NORM
    assert pointwise(got, ref, s) < TOL_POINT
    s2, xi = cp.PowerToCorrelation(k)(rows)                   # the facade adds nothing: same numbers
    np.testing.assert_array_equal(xi, got)
    np.testing.assert_array_equal(cp.PowerToCorrelation(k, rescale_rows=True)(np.zeros((2, k.size)))[1], 0.)   # accepted, now the default


@pytest.mark.parametrize('n', [5000, 8192, 20000, 70000, 300000])
def test_large_sizes(cp, n):
    """Padded sizes beyond the LDS-resident kernel (Np = 16384, 16384, 65536, 262144, 1048576: column transforms of 4, 16, 64 and 256 points
    around the 4096-point row kernel, csrc/cp_fftlog_large.hip) follow the same reference arithmetic: every extrap mode, keep_padding,
    multi-kernel, odd batch."""
    rng = np.random.default_rng(n)
    k = np.logspace(-4, 2, n)
    t = ofl.power_to_correlation(k, ell=[0, 2])
    f = cp.PowerToCorrelation(k, ell=[0, 2])
    assert f.padded_size > 8192
    fun = rng.uniform(0.95, 1.05, size=(3, 2, n)) * k**-1.2
    with np.errstate(all='ignore'):
        modes = [(0, False), (0, True), ('edge', False), ((1.5, 'edge'), False)]
        if n <= 8192:    # the geometric continuation of noisy rows over > 20000 padded samples overflows (in the reference as well)
            modes += [('log', True), (('log', 0.3), False)]
        for extrap, keep in modes:
            ref = ofl.apply(t, fun, extrap=extrap, keep_padding=keep)
            y, got = f(fun, extrap=extrap, keep_padding=keep)
            assert got.shape == ref.shape
            scale = np.abs(ofl.pad(fun, (t.in_left, t.in_right), extrap) * t.pre).max(axis=-1)
            post = np.abs(t.post[:, t.out_left:t.out_left + t.n] if not keep else t.post)
            err = np.abs(got - ref) / post / scale[..., None]
            assert err.max() < 1e-13, (n, extrap, keep, err.max())
    s, xi = cp.PowerToCorrelation(k, ell=0)(fun[0, 0])
    assert xi.shape == (n,) and tilted_err(xi, ofl.apply(ofl.power_to_correlation(k, ell=0), fun[0, 0])[0], s, 1.5) < TOL_NORM


def test_complex_transforms_and_their_inverse(cp, golden):
    """complex=True multipoles (reference fftlog.py:318-330): xi_ell x (-i)^ell as complex128; inv() of such a plan (fftlog.py:243-248) has a
    complex prefactor -- the FFT then sees the real part of fun x prefactor, as numpy.fft.rfft does -- and brings P(k) back."""
    pkd = golden('pk_eh_default')
    k, pk = pkd['k1024'], pkd['pk1024']
    ells = [0, 1, 2, 3]
    f = cp.PowerToCorrelation(k, ell=ells, complex=True)
    s, xi = f(pk)
    assert xi.dtype == np.complex128 and xi.shape == (4, 1024)
    real = cp.PowerToCorrelation(k, ell=ells)(pk)[1]
    for i, ell in enumerate(ells):      # (-i)^ell against the real transform's (-1)^(ell // 2)
        np.testing.assert_allclose(xi[i], real[i] * (-1j)**ell / (-1.)**(ell // 2), rtol=1e-13, atol=0)
    t = ofl.power_to_correlation(k, ell=ells)
    ref = ofl.apply(t, pk)
    for i in range(4):
        assert tilted_err(real[i], ref[i], s[i], 1.5) < TOL_NORM
    f.inv()
    assert np.iscomplexobj(f.padded_prefactor)
    k2, pk2 = f(xi)                       # complex in, real table x unit phase on the way in
    m = (k2[0] > 1e-2) & (k2[0] < 10.)
    for i in range(4):
        np.testing.assert_allclose(pk2[i][m].real, pk[m], rtol=2e-3)
    # the same inverse from the real convention: identical numbers
    g = cp.PowerToCorrelation(k, ell=ells)
    g.inv()
    np.testing.assert_allclose(g(real)[1][:, m], pk2[:, m].real, rtol=1e-9)


def test_nonfinite_rows_stay_isolated(cp, golden):
    """A NaN / Inf row gives a NaN row and leaves its neighbours alone, as numpy's row-by-row FFT does (the kernel packs rows in pairs)."""
    pkd = golden('pk_eh_default')
    k, pk = pkd['k1024'], pkd['pk1024']
    rows = np.tile(pk, (5, 1))
    rows[1, 17] = np.nan
    rows[2, 900] = np.inf
    f = cp.PowerToCorrelation(k)
    s, xi = f(rows)
    ref = ofl.apply(ofl.power_to_correlation(k), pk)[0]
    assert np.isnan(xi[1]).all() and np.isnan(xi[2]).all()
    for i in (0, 3, 4):
        assert tilted_err(xi[i], ref, s, 1.5) < TOL_NORM
    import torch
    t = torch.as_tensor(rows, device='cuda')
    xi_t = f(t)[1]
    assert bool(torch.isnan(xi_t[1]).all()) and bool(torch.isfinite(xi_t[0]).all())


@pytest.mark.parametrize('size', [8, 16, 32, 64, 256, 512, 1024, 4096, 8192, 16384])
def test_engine_forward_backward_are_real_ffts(cp, size):
    """NumpyFFTEngine / FFTWEngine by name (reference fftlog.py:533-544): forward = rfft, backward = irfft(conj(.), n=size) on the package's own
    device FFTs (cp_rfft_forward / cp_rfft_backward) against numpy's, rows of very different magnitudes each on its own scale (one transform per
    row), leading axes kept, numpy in -> numpy out and tensor in -> tensor out."""
    import torch
    from cosmoprimo_amd.fftlog import NumpyFFTEngine, FFTWEngine
    rng = np.random.default_rng(size)
    x = rng.normal(size=(3, 5, size)) * np.array([1e-12, 1., 1e9])[:, None, None]
    for engine in (NumpyFFTEngine(size), FFTWEngine(size, nparallel=5, plan='estimate')):
        spectrum = engine.forward(x)
        ref = np.fft.rfft(x, axis=-1)
        assert isinstance(spectrum, np.ndarray) and spectrum.dtype == np.complex128 and spectrum.shape == (3, 5, size // 2 + 1)
        for i in range(3):
            assert np.abs(spectrum[i] - ref[i]).max() < 4e-16 * np.log2(size) * np.abs(ref[i]).max() * 4
        assert (spectrum[..., 0].imag == 0.).all() and (spectrum[..., -1].imag == 0.).all()
        z = rng.normal(size=(3, 5, size // 2 + 1)) + 1j * rng.normal(size=(3, 5, size // 2 + 1))
        z *= np.array([1e-12, 1., 1e9])[:, None, None]
        back = engine.backward(z)
        ref = np.fft.irfft(z.conj(), n=size, axis=-1)
        assert back.dtype == np.float64 and back.shape == (3, 5, size)
        for i in range(3):
            assert np.abs(back[i] - ref[i]).max() < 4e-16 * np.log2(size) * np.abs(ref[i]).max() * 4
        there_and_back = engine.backward(engine.forward(x).conj())
        for i in range(3):
            assert np.abs(there_and_back[i] - x[i]).max() < 1e-14 * np.abs(x[i]).max()
        t = engine.forward(torch.as_tensor(x[1, 0], device='cuda'))
        assert t.is_cuda and t.dtype == torch.complex128 and t.shape == (size // 2 + 1,)
        np.testing.assert_array_equal(t.cpu().numpy(), spectrum[1, 0])
    bad = x[1].copy()
    bad[2, 3] = np.nan
    out = NumpyFFTEngine(size).forward(bad)
    assert np.isnan(out[2]).all() and np.isfinite(np.delete(out, 2, axis=0)).all()
    with pytest.raises(ValueError):
        NumpyFFTEngine(size).forward(np.zeros(size + 1))


@pytest.mark.parametrize('size', [6, 24, 100, 255, 1000, 8190])
def test_engines_of_any_size(cp, size):
    """Sizes the row kernels do not take (anything but a power of two from 8 to 16 384: numpy.fft.rfft takes any, reference fftlog.py:533-544) go through
    Bluestein's algorithm on the package's own transforms: rows against numpy, each on its own scale; odd and even sizes."""
    import torch
    from cosmoprimo_amd.fftlog import NumpyFFTEngine
    rng = np.random.default_rng(size)
    x = rng.normal(size=(2, 3, size)) * np.array([1e-9, 1e6])[:, None, None]
    engine = NumpyFFTEngine(size)
    spectrum, ref = engine.forward(x), np.fft.rfft(x, axis=-1)
    assert spectrum.dtype == np.complex128 and spectrum.shape == ref.shape
    for i in range(2):
        assert np.abs(spectrum[i] - ref[i]).max() < 1e-13 * np.abs(ref[i]).max()
    z = rng.normal(size=(2, 3, size // 2 + 1)) + 1j * rng.normal(size=(2, 3, size // 2 + 1))
    back, ref = engine.backward(z), np.fft.irfft(z.conj(), n=size, axis=-1)
    assert back.dtype == np.float64 and back.shape == ref.shape
    assert np.abs(back - ref).max() < 1e-13 * np.abs(ref).max()
    assert np.abs(engine.backward(engine.forward(x[1]).conj()) - x[1]).max() < 1e-13 * np.abs(x[1]).max()
    t = engine.forward(torch.as_tensor(x[1, 0], device='cuda'))
    assert t.is_cuda and t.shape == (size // 2 + 1,)
    with pytest.raises(NotImplementedError):      # beyond what a convolution of 16 384 samples covers
        NumpyFFTEngine(3 * 4096).forward(np.zeros(3 * 4096))


def test_engine_instance_gives_the_fused_result(cp, golden):
    """FFTlog(engine=NumpyFFTEngine(size)) is the fused kernel: bit-identical to the default engine; the same class wrapped in a foreign object
    (only forward / backward visible: the reference's plug point) runs un-fused around the device FFTs and agrees to 1e-13 in tilted space."""
    from cosmoprimo_amd.fftlog import NumpyFFTEngine
    pkd = golden('pk_eh_default')
    k, pk = pkd['k1024'], pkd['pk1024']
    rows = np.stack([pk, 3. * pk, pk**1.01])
    s0, ref = cp.PowerToCorrelation(k, ell=0)(rows)
    s1, got = cp.PowerToCorrelation(k, ell=0, engine=NumpyFFTEngine(2048))(rows)
    np.testing.assert_array_equal(s1, s0)
    np.testing.assert_array_equal(got, ref)

    class Foreign(object):
        def __init__(self, size):
            self._inner, self.size = NumpyFFTEngine(size), size

        def forward(self, fun):
            return self._inner.forward(fun)

        def backward(self, fun):
            return self._inner.backward(fun)

    s2, unfused = cp.PowerToCorrelation(k, ell=0, engine=Foreign(2048))(rows)
    for a, b in zip(unfused, ref):
        assert tilted_err(a, b, s0, 1.5) < TOL_NORM


def test_result_buffer_of_the_caller(cp):
    """``FFTlog.__call__(out=)``: the kernel writes the caller's tensor (no allocation per call), the result is a view of it; wrong buffers are refused."""
    import torch
    n = 256
    k = np.logspace(-3, 2, n)
    rows = torch.as_tensor(np.tile(k**-1.5, (9, 1)) * np.linspace(1., 2., 9)[:, None], device='cuda')
    f1 = cp.PowerToCorrelation(k, ell=0)
    s, ref = f1(rows)
    buf = torch.empty(9 * n, dtype=torch.float64, device='cuda')
    s2, got = f1(rows, out=buf)
    assert got.data_ptr() == buf.data_ptr() and got.shape == (9, n) and torch.equal(got, ref)
    f3 = cp.PowerToCorrelation(k, ell=[0, 2, 4])
    buf3 = torch.empty((9, 3, n), dtype=torch.float64, device='cuda')
    assert torch.equal(f3(rows[:, None, :], out=buf3)[1], f3(rows[:, None, :])[1])
    for bad in (torch.empty(9 * n - 1, dtype=torch.float64, device='cuda'), torch.empty(9 * n, dtype=torch.float32, device='cuda'),
                torch.empty(9 * n, dtype=torch.float64), torch.empty((9, 2 * n), dtype=torch.float64, device='cuda')[:, ::2]):
        with pytest.raises(ValueError):
            f1(rows, out=bad)
    with pytest.raises(ValueError):
        cp.PowerToCorrelation(k, ell=2, complex=True)(rows, out=buf)


def test_two_host_threads_on_one_plan(cp):
    """execute is thread-safe across distinct streams for one plan (read-only tables).  Two host threads, each with its own stream,
    its own rows and its own output, hammer ONE cp_fftlog_plan through the C ABI at the same time; every result equals the one-thread result bit for bit."""
    import threading
    import torch
    from cosmoprimo_amd import _lib
    n, nb, reps = 2048, 4096, 12
    k = np.logspace(-5, 2, n)
    f = cp.PowerToCorrelation(k, ell=0)
    dev = torch.device('cuda', torch.cuda.current_device())
    plan, lib = f._get_plan(dev), _lib.load()
    rng = np.random.default_rng(11)
    inputs = [torch.as_tensor(rng.uniform(0.5, 2., (nb, 1)) * k**-1.2, device=dev) for _ in range(2)]
    expected = [f(rows)[1] for rows in inputs]
    torch.cuda.synchronize()
    outs = [[torch.empty_like(rows) for _ in range(reps)] for rows in inputs]
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    errors = []
    start = threading.Barrier(2)

    def work(i):
        try:
            start.wait()
            for r in range(reps):
                _lib.check(lib.cp_fftlog_execute(plan.handle, inputs[i].data_ptr(), outs[i][r].data_ptr(), nb, 0, 0., 0, 0., 0, streams[i].cuda_stream))
            streams[i].synchronize()
        except Exception as exc:      # noqa: BLE001
            errors.append(exc)

    threads = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    for i in range(2):
        for r in range(reps):
            assert torch.equal(outs[i][r], expected[i]), (i, r)
