"""Dense linear operators (``cp_linop_plan_create`` + ``cp_spline_apply``): the float64 matrix-core kernel (v_mfma_f64_16x16x4_f64) against the
vector-ALU kernel and against numpy, for shapes that exercise every edge of the 32 x 256 tiling; the fused outer-product epilogue of the spline
operator (``cp_spline_apply_outer``) against its two-step form."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('n,nq,nrows', [(16, 1, 1), (17, 64, 16), (100, 65, 33), (341, 341, 1000), (1024, 256, 4097), (1000, 1030, 64)])
def test_dense_operator_paths(n, nq, nrows):
    import torch
    from cosmoprimo_amd.spline import LinearOperator
    rng = np.random.default_rng(n + nq)
    w = rng.normal(size=(nq, n)) / np.sqrt(n)
    if nq > 3:
        w[3] = np.nan                     # a query marked as outside: NaN out
    y = rng.normal(size=(nrows, n))
    op = LinearOperator.dense(w)
    ty = torch.as_tensor(y, device=op.device)
    ref = y.dot(w.T)
    scale = np.abs(y).max() * np.nanmax(np.abs(w)) * n
    for path in ('valu', 'mfma', None):
        got = op(ty, path=path).cpu().numpy()
        assert got.shape == ref.shape and np.array_equal(np.isnan(got), np.isnan(ref)), path
        assert np.nanmax(np.abs(got - ref)) < 1e-14 * scale, (path, np.nanmax(np.abs(got - ref)) / scale)
    both = op(ty.abs(), sqrt=True, scale=2., path='mfma').cpu().numpy()
    ref2 = np.sqrt(2. * np.abs(y).dot(w.T))
    keep = np.isfinite(ref2) & np.isfinite(both)       # (sums that cancel to ~0 may fall on either side of it)
    np.testing.assert_allclose(both[keep]**2, ref2[keep]**2, rtol=0., atol=4e-14 * scale)
    # a NaN row of Y stays in its row
    y2 = y.copy()
    y2[nrows // 2, n - 1] = np.nan
    got = op(torch.as_tensor(y2, device=op.device), path='mfma').cpu().numpy()
    cols = [c for c in range(nq) if c != 3]
    bad = np.isnan(got[:, cols]).any(axis=1)
    assert bad[nrows // 2] and bad.sum() == 1


def test_large_vector_route_plan_keeps_no_dense_copy():
    """A spline from many more knots than queries runs on the vector route (the window of knots under 64 queries is many bandwidths wide); its
    dense copy (16 384 x 1024 doubles = 134 MB per plan) is not built: forcing the matrix-core route is refused, the default and the vector
    route agree with the dense operator."""
    import torch
    from cosmoprimo_amd.spline import LinearOperator, dense_operator
    rng = np.random.default_rng(5)
    x, xq = np.linspace(0., 1., 16384), np.sort(rng.uniform(0., 1., 1024))
    LinearOperator.spline(x[:64], xq[:8] * x[63], bc='natural')      # (the library and its kernels are loaded)
    torch.cuda.synchronize()
    before = torch.cuda.mem_get_info()[0]
    op = LinearOperator.spline(x, xq, bc='natural')
    assert before - torch.cuda.mem_get_info()[0] < (64 << 20)
    y = rng.normal(size=(20, 16384))
    ty = torch.as_tensor(y, device=op.device)
    with pytest.raises(ValueError):
        op(ty, path='mfma')
    ref = y.dot(dense_operator(x, xq, bc='natural').T)
    for path in (None, 'valu'):
        assert np.abs(op(ty, path=path).cpu().numpy() - ref).max() < 1e-13 * np.abs(ref).max()


@pytest.mark.parametrize('n,nq,bc,nrows', [(512, 300, 'natural', 1), (504, 1024, 'not-a-knot', 77), (2048, 2048, 'clamped', 40), (1024, 256, 'natural', 33),
                                           (40, 1000, 'not-a-knot', 16), (3666, 1024, 'clamped', 19)])
def test_banded_operator_on_both_paths(n, nq, bc, nrows):
    """Spline operators keep a dense copy as well and run on the matrix cores as a block-banded GEMM (a tile of 64 queries times the window of
    knots under its bands): same numbers as the banded vector kernel, whichever of the two the library picks, NaN queries outside the knots."""
    import torch
    from cosmoprimo_amd.spline import LinearOperator, dense_operator
    rng = np.random.default_rng(n + nq)
    x = np.sort(rng.uniform(0., 1., n)) if bc != 'clamped' else np.linspace(0., 1., n)
    xq = np.concatenate([[-0.1], np.sort(rng.uniform(x[0], x[-1], nq - 2)), [1.2]])
    op = LinearOperator.spline(x, xq, bc=bc, nu=2 if bc == 'clamped' else 0)
    y = rng.normal(size=(nrows, n))
    ty = torch.as_tensor(y, device=op.device)
    w = dense_operator(x, xq, bc=bc, nu=2 if bc == 'clamped' else 0)
    ref = y.dot(np.nan_to_num(w).T)
    scale = np.abs(y).dot(np.abs(np.nan_to_num(w)).T).max()
    results = {path: op(ty, path=path).cpu().numpy() for path in ('valu', 'mfma', None)}
    for path, got in results.items():
        assert np.isnan(got[:, 0]).all() and np.isnan(got[:, -1]).all() and np.isfinite(got[:, 1:-1]).all(), path
        assert np.abs(got[:, 1:-1] - ref[:, 1:-1]).max() < 4e-15 * scale, (path, np.abs(got[:, 1:-1] - ref[:, 1:-1]).max() / scale)
    assert np.array_equal(results[None], results['valu'], equal_nan=True) or np.array_equal(results[None], results['mfma'], equal_nan=True)


def test_outer_epilogue():
    import torch
    from cosmoprimo_amd.spline import LinearOperator
    rng = np.random.default_rng(3)
    x = np.geomspace(1e-2, 1e2, 1024)
    xq = np.geomspace(0.1, 50., 256)
    op = LinearOperator.spline(np.log(x), np.log(xq), bc='natural')
    for nrows, nz in [(1, 1), (5, 64), (37, 7), (200, 30)]:
        y = torch.as_tensor(rng.uniform(0.5, 2., size=(nrows, 1024)) * x**-0.7, device=op.device)
        g = torch.as_tensor(rng.uniform(0.1, 1., size=(nrows, nz)), device=op.device)
        two_steps = (op(y)[:, :, None] * g[:, None, :]).sqrt()
        fused = op.outer(y, g, sqrt=True)
        assert fused.shape == (nrows, 256, nz)
        assert torch.equal(fused, two_steps) or float(((fused - two_steps) / two_steps).abs().max()) < 1e-14
        plain = op.outer(y, g, scale=3.)
        np.testing.assert_allclose(plain.cpu().numpy(), (3. * op(y)[:, :, None] * g[:, None, :]).cpu().numpy(), rtol=1e-15)
    outside = LinearOperator.spline(np.log(x), np.log(np.array([1e-3, 1., 1e3])), bc='natural')
    res = outside.outer(y[:2], g[:2])
    assert bool(torch.isnan(res[:, 0]).all()) and bool(torch.isnan(res[:, 2]).all()) and bool(torch.isfinite(res[:, 1]).all())


@pytest.mark.parametrize('n,nq,m,nb', [(5, 1, 1, 1), (30, 64, 1024, 7), (33, 70, 300, 3), (16, 130, 257, 2), (30, 64, 64, 100)])
def test_dense_operator_along_the_middle_axis(n, nq, m, nb):
    """``cp_linop_apply_mid``: out[b, q, c] = post(scale sum_j W[q, j] y[b, j, c]) against numpy, every edge of the 64 x 256 tiling, the three
    epilogues, NaN queries, and a NaN in one batch entry (or one column) staying where it is."""
    import torch
    from cosmoprimo_amd.spline import LinearOperator
    rng = np.random.default_rng(n + nq + m)
    w = rng.normal(size=(nq, n)) / np.sqrt(n)
    if nq > 3:
        w[3] = np.nan
    y = rng.normal(size=(nb, n, m))
    op = LinearOperator.dense(w)
    ty = torch.as_tensor(y, device=op.device)
    ref = np.einsum('qj,bjc->bqc', w, y)
    scale = np.abs(y).max() * np.nanmax(np.abs(w)) * n
    got = op.mid(ty).cpu().numpy()
    assert got.shape == ref.shape and np.array_equal(np.isnan(got), np.isnan(ref))
    assert np.nanmax(np.abs(got - ref)) < 1e-14 * scale
    np.testing.assert_allclose(op.mid(ty, post='exp10', scale=0.5).cpu().numpy(), 10**(0.5 * ref), rtol=1e-13)
    both = op.mid(ty.abs(), post='sqrt', scale=2.).cpu().numpy()
    ref2 = 2. * np.einsum('qj,bjc->bqc', w, np.abs(y))
    keep = np.isfinite(both) & (ref2 > 0)
    np.testing.assert_allclose(both[keep]**2, ref2[keep], rtol=0., atol=4e-14 * scale)
    y2 = y.copy()
    y2[nb // 2, n - 1, m // 2] = np.nan
    got = op.mid(torch.as_tensor(y2, device=op.device)).cpu().numpy()
    rows = [q for q in range(nq) if q != 3]
    bad = np.isnan(got[:, rows])
    assert bad[nb // 2, :, m // 2].all() and bad.sum() == len(rows)
    with pytest.raises(ValueError):
        op.mid(ty[:, :-1])
    # a spline plan along the middle axis
    from cosmoprimo_amd.spline import dense_operator
    xk, xq = np.linspace(0., 1., 40), np.linspace(0., 1., 7)
    z = rng.normal(size=(2, 40, 9))
    got = LinearOperator.spline(xk, xq).mid(torch.as_tensor(z, device=op.device)).cpu().numpy()
    np.testing.assert_allclose(got, np.einsum('qj,bjc->bqc', dense_operator(xk, xq), z), rtol=0., atol=1e-13)


def test_exp10_epilogue_over_the_whole_line():
    """The branch-free 10^x of the operator epilogues (csrc/cp_math.h: exp10_mid) where the result leaves the doubles: subnormal results, the
    underflow to 0 (never a negative or huge value from a polynomial evaluated outside its range), the overflow to Inf, NaN and +-Inf."""
    import torch
    from cosmoprimo_amd.spline import LinearOperator
    x = np.concatenate([np.linspace(-420., -300., 1201), np.linspace(300., 320., 201), np.linspace(-30., 30., 61),
                        [-1e300, -1100., -400.5, -331.2, -324.1, -323.9, -323.3, -308., -307.6, 308.25, 308.2548, 308.31, 400.5, 1100., 1e300,
                         np.nan, np.inf, -np.inf]])
    op = LinearOperator.dense(np.array([[1.]]))
    got = op.mid(torch.as_tensor(x[None, None, :], device=op.device), post='exp10').cpu().numpy().ravel()
    with np.errstate(over='ignore', under='ignore'):
        ref = np.power(10., x)
    assert np.array_equal(np.isnan(got), np.isnan(ref))
    assert (got[~np.isnan(got)] >= 0.).all()
    normal = np.isfinite(ref) & (ref > 2.3e-308)
    np.testing.assert_allclose(got[normal], ref[normal], rtol=1e-15 * 4)
    sub = np.isfinite(ref) & (ref <= 2.3e-308)
    assert np.abs(got[sub] - ref[sub]).max() <= 2 * 4.95e-324      # subnormals: within two units of the last place of the subnormal range
    assert np.array_equal(np.isinf(got), np.isinf(ref))
