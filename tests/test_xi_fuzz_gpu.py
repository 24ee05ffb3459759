"""GPU parity of the tabulated xi(s) / xi(s, z) interpolators with options drawn at random (separations geometric or jittered, interpolation in s or
log s, spline degrees, growth factor or table in z): values inside, AT the ends of the table and outside (NaN), pairs, ignore_growth, to_1d, to_pk and
sigma8 through it -- NaN where the transformed P(k) goes negative and the log-log interpolator refuses it, in the reference's own pattern (everywhere for
cubic splines, at and below the offending knot for a (k, z) table interpolated linearly along k: FITPACK's elimination order,
tests/test_fitpack_nan_host.py) -- against the reference's own outputs
(tests/golden/xi_fuzz.npz, `python -m oracle.gen_golden xi_fuzz`)."""
import warnings

import numpy as np
import pytest

from oracle.gen_golden import xi_fuzz_configs, xi_fuzz_outputs, XI_FUZZ_N

pytestmark = pytest.mark.gpu
RTOL = {'xi': 1e-10, 'xi_pairs': 1e-10, 'xi_nogrowth': 1e-10, 'to_1d': 1e-10, 'pk': 1e-8, 'sigma8': 1e-8, 'sigma8_z': 1e-8}


@pytest.mark.parametrize('i', range(XI_FUZZ_N))
def test_random_options(golden, i):
    import torch
    assert torch.cuda.is_available()
    import cosmoprimo_amd as cp
    warnings.simplefilter('ignore')
    g = golden('xi_fuzz')
    cfg = xi_fuzz_configs()[i]
    got = xi_fuzz_outputs(cp, cfg)
    names = [key[len('c%d_' % i):] for key in g if key.startswith('c%d_' % i)]
    assert sorted(got) == sorted(names), cfg
    for name in names:
        ref = g['c%d_%s' % (i, name)]
        assert got[name].dtype.kind == 'f' and ref.dtype.kind == 'f', (cfg, name, str(got[name]), str(ref))
        assert got[name].shape == ref.shape, (cfg, name, got[name].shape, ref.shape)
        assert np.array_equal(np.isnan(got[name]), np.isnan(ref)), (cfg, name)
        scale = np.nanmax(np.abs(ref)) if np.isfinite(ref).any() else 1.
        # xi changes sign and two rows share a transform: rounding relative to the scale of the array
        np.testing.assert_allclose(got[name], ref, rtol=RTOL[name], atol=1e-13 * scale, equal_nan=True, err_msg='%s of %s' % (name, cfg))
