"""The FFTLog oracle on configurations drawn at random -- class, size (powers of two and not), range, tilt, folds, low-ringing, xy, padding mode,
several transforms at once, batches, complex phases -- against the reference's own outputs for them (tests/golden/fftlog_fuzz.npz,
`python -m oracle.gen_golden fftlog_fuzz`)."""
import numpy as np
import pytest

from oracle import fftlog as of
from oracle.gen_golden import fftlog_fuzz_configs, fftlog_fuzz_stride, fftlog_fuzz_error, FFTLOG_FUZZ_N

KERNELS = {'SphericalBesselJKernel': of.u_spherical_bessel_j, 'TophatKernel': of.u_tophat, 'TophatSqKernel': of.u_tophat_sq, 'BesselJKernel': of.u_bessel_j}


def oracle_transform(cfg):
    """(y, g) of a configuration through oracle/fftlog.py, in the reference's output conventions."""
    x = np.logspace(cfg['lo'], cfg['lo'] + cfg['span'], cfg['n'])
    kw = dict(minfolds=cfg['minfolds'], lowring=cfg['lowring'], xy=cfg['xy'])
    kind = cfg['kind']
    if kind == 'PowerToCorrelation':
        t = of.power_to_correlation(x, ell=cfg['ell'], q=cfg['q'], **kw)
        if cfg['complex']:      # (-i)^ell instead of (-1)^(ell // 2) (fftlog.py:326-330)
            ells = np.atleast_1d(cfg['ell'])
            t.post = t.post * ((-1j)**ells / (-1.)**(ells // 2))[:, None]
    elif kind == 'CorrelationToPower':
        t = of.correlation_to_power(x, ell=cfg['ell'], q=cfg['q'], **kw)
    elif kind == 'TophatVariance':
        t = of.tophat_variance(x, q=cfg['q'], **kw)
    elif kind == 'GaussianVariance':
        t = of.gaussian_variance(x, q=cfg['q'], **kw)
    elif kind == 'HankelTransform':
        t = of.hankel(x, nu=cfg['nu'], q=cfg['q'], **kw)
    else:
        name, arg = cfg['kernel']
        u = of.u_gaussian if name == 'GaussianKernel' else (lambda z: KERNELS[name](z, arg))
        t = of.setup(x[None, :], [u], [cfg['q']], **kw)
    xm = 10.**(cfg['lo'] + cfg['knee'] * cfg['span'])
    fun = (x / xm)**cfg['slope'] / (1. + (x / xm)**2)**1.5
    if cfg['nbatch']:
        fun = fun * np.array([1., 0.5, 2.5])[:cfg['nbatch'], None] * (x / xm)**(0.1 * np.arange(cfg['nbatch'])[:, None])
        if np.ndim(cfg.get('ell', 0)):
            fun = fun[:, None, :] * np.ones((1, len(cfg['ell']), 1))
    g = of.apply(t, fun, extrap=cfg['extrap'], keep_padding=cfg['keep_padding'])
    y = t.padded_y if cfg['keep_padding'] else t.y
    if t.nker == 1:      # one transform: y is 1-D and g has the shape of the input (fftlog.py:236-241)
        y = y[0]
        g = g.reshape(np.shape(fun)[:-1] + (g.shape[-1],))
    return y, g


@pytest.mark.parametrize('i', range(FFTLOG_FUZZ_N))
def test_random_configurations(golden, i):
    g = golden('fftlog_fuzz')
    cfg = fftlog_fuzz_configs()[i]
    y, out = oracle_transform(cfg)
    stride = fftlog_fuzz_stride(int(g['c%d_size' % i]))
    assert y.shape[-1] == int(g['c%d_size' % i]), cfg
    y, out = y[..., ::stride], out[..., ::stride]
    ref_y, ref = g['c%d_y' % i], g['c%d_g' % i]
    assert y.shape == ref_y.shape and out.shape == ref.shape, (cfg, y.shape, ref_y.shape, out.shape, ref.shape)
    np.testing.assert_allclose(y, ref_y, rtol=1e-13, err_msg=str(cfg))
    # norm-wise in the tilted space g y^q, where the transform's rounding is uniform
    # norm-wise in the tilted space g y^q, where the transform's rounding is uniform; same FFT as the reference: no allowance needed
    assert fftlog_fuzz_error(cfg, out, ref, ref_y) <= 1e-12, cfg
