"""GPU parity of the FFTLog classes at sizes beyond the LDS-resident kernel (padded lengths 16 384 ... 131 072: the four-step path of
csrc/cp_fftlog_large.hip) on random configurations -- class / kernel, range, tilt, folds, low-ringing or xy, padding mode, several ell, batches,
keep_padding -- against the reference's own outputs (tests/golden/fftlog_large.npz, `python -m oracle.gen_golden fftlog_large`): output coordinates
1e-13, transforms norm-wise 1e-12 in the tilted space (+ MOVES_FACTOR = 8 x the reference's own movement under one-ulp inputs where the padding makes the problem
ill-conditioned)."""
import numpy as np
import pytest

from oracle.gen_golden import fftlog_large_configs, fftlog_fuzz_build, fftlog_fuzz_stride, fftlog_fuzz_error

pytestmark = pytest.mark.gpu
# error / movement of the reference's own result, measured over the configurations above 1e-12 (profiles/r6_fftlog_fuzz_errors.txt, tools/fuzz_error_distribution.py):
# 11 of 60 + 3 of 12 configurations, median 1.0, maximum 3.5 -- the bound is twice the maximum and a bit (it was an unargued 30 until round 6)
MOVES_FACTOR = 8.


@pytest.mark.parametrize('i', range(12))
def test_large_sizes(golden, i):
    import torch
    assert torch.cuda.is_available()
    from cosmoprimo_amd import fftlog as fl
    g = golden('fftlog_large')
    cfg = fftlog_large_configs()[i]
    obj, x, fun = fftlog_fuzz_build(fl, cfg)
    assert obj.padded_size > 8192
    y, out = obj(fun, extrap=cfg['extrap'], keep_padding=cfg['keep_padding'])
    y, out = np.asarray(y), np.asarray(out)
    assert y.shape[-1] == int(g['c%d_size' % i]), cfg
    stride = fftlog_fuzz_stride(y.shape[-1])
    y, out = y[..., ::stride], out[..., ::stride]
    ref_y, ref = g['c%d_y' % i], g['c%d_g' % i]
    assert y.shape == ref_y.shape and out.shape == ref.shape and out.dtype == ref.dtype, (cfg, y.shape, ref_y.shape, out.shape, ref.shape, out.dtype)
    np.testing.assert_allclose(y, ref_y, rtol=1e-13, err_msg=str(cfg))
    err = fftlog_fuzz_error(cfg, out, ref, ref_y)
    assert err <= 1e-12 + MOVES_FACTOR * float(g['c%d_moves' % i]), (cfg, err, float(g['c%d_moves' % i]))
