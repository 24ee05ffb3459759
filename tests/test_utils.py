"""LeastSquareSolver (host numpy): the reference's own known-answer identities (tests/test_utils.py:8-60)
and agreement with the oracle's restatement."""
import numpy as np

from oracle import bao as obao


def test_least_squares():
    from cosmoprimo_amd.utils import LeastSquareSolver
    for compute_inverse in [False, True]:
        x = np.linspace(1, 100, 10)
        gradient = np.array([1. / x, np.ones_like(x), x, x ** 2, x ** 3])
        rng = np.random.RandomState(seed=42)
        y = rng.uniform(0., 1., x.size)
        for cov in [np.diag(x), np.diag(x) + 0.1]:
            precision = np.linalg.inv(cov)
            solver = LeastSquareSolver(gradient, precision, compute_inverse=compute_inverse)
            result = solver(y)
            # normal equations: gradient F (y - p gradient) = 0
            np.testing.assert_allclose(gradient.dot(precision).dot(y - result.dot(gradient)), 0., atol=1e-8)
            lss_c = LeastSquareSolver(gradient, precision, constraint_gradient=np.ones((len(gradient), 1)), compute_inverse=compute_inverse)
            result = lss_c(y, constraint=0.42)
            assert lss_c.chi2() >= solver.chi2()
            assert np.allclose(sum(result), 0.42)
            weights = np.arange(len(gradient))
            lss_c = LeastSquareSolver(gradient, precision, constraint_gradient=np.column_stack([np.ones(len(gradient)), weights]), compute_inverse=compute_inverse)
            result = lss_c(y, constraint=[0.42, 2.])
            assert lss_c.chi2() >= solver.chi2()
            assert np.allclose(sum(result), 0.42) and np.allclose(sum(r * w for r, w in zip(result, weights)), 2.)
        result_ref = LeastSquareSolver(gradient, precision=np.eye(x.size), compute_inverse=compute_inverse)(y)
        for precision in [1., np.ones_like(x)]:
            assert np.allclose(LeastSquareSolver(gradient, precision=precision, compute_inverse=compute_inverse)(y), result_ref)
        solver = LeastSquareSolver(gradient, precision=np.eye(x.size), compute_inverse=compute_inverse)
        ys = np.array([y] * 12)
        result = solver(ys)
        assert result.shape == (len(ys), len(gradient)) and np.allclose(result, result_ref)
        assert solver.model().shape == ys.shape and solver.chi2().shape == (len(ys),)
        solver = LeastSquareSolver(np.ones_like(x), precision=np.eye(x.size), compute_inverse=compute_inverse)
        assert solver(y).ndim == 0 and solver(ys).shape == (len(ys),)
    # the constrained fit of the BAO filters: same model as the oracle's restatement
    k = np.geomspace(1e-3, 1., 50)
    gradient = np.array([k**(i - 1) for i in range(4)])
    cg = np.column_stack([gradient[..., 0], gradient[..., 1] - gradient[..., 0], gradient[..., -1], gradient[..., -2] - gradient[..., -1]])
    d = 1. + 0.05 * np.sin(40. * k)
    c = [d[0], d[1] - d[0], d[-1], d[-2] - d[-1]]
    s = LeastSquareSolver(gradient, precision=k**2, constraint_gradient=cg, compute_inverse=False)
    s(d, constraint=c)
    np.testing.assert_allclose(s.model(), obao.least_squares_constrained(gradient, k**2, cg, d, c), rtol=1e-12)
    np.testing.assert_allclose(s.model()[[0, -1]], d[[0, -1]], rtol=1e-10)
