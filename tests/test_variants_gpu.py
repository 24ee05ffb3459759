"""GPU parity of the eisenstein_hu_nowiggle_variants engine (cp_power_eval_variants) against golden vectors from
the reference (tests/golden/variants.npz) and the oracle: scalars 1e-13, transfer 1e-11, P(k, z) and sigma8 1e-10."""
import warnings

import numpy as np
import pytest

from oracle import power as op
from oracle.gen_golden import VARIANTS_PARAMS
from test_oracle_variants import NAMES, background_params

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def cp():
    import torch
    assert torch.cuda.is_available(), 'GPU tests need a ROCm device'
    import cosmoprimo_amd
    return cosmoprimo_amd


@pytest.mark.parametrize('ic', range(len(VARIANTS_PARAMS)))
def test_variants(cp, golden, ic):
    g = golden('variants')
    k, z = g['k'], g['z']
    pre = 'c%d_' % ic
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        cosmo = cp.Cosmology(engine='eisenstein_hu_nowiggle_variants', **VARIANTS_PARAMS[ic])
        eng = cosmo.engine
        for name in NAMES:
            np.testing.assert_allclose(getattr(eng, name), g[pre + name], rtol=1e-13, err_msg=name)
        tr, fo, ba = cosmo.get_transfer(), cosmo.get_fourier(), cosmo.get_background()
        np.testing.assert_allclose(ba.growth_factor(z, znorm=eng.z_eq), g[pre + 'growth_k0'], rtol=1e-11)
        for of in ['delta_m', 'delta_cb']:
            t = tr.transfer_kz(k, z, of=of)
            assert t.shape == (k.size, z.size)
            np.testing.assert_allclose(t, g[pre + 'transfer_' + of], rtol=1e-11, err_msg=of)
            np.testing.assert_allclose(t, op.variants_transfer_kz(k, z, background_params(VARIANTS_PARAMS[ic]), op.variants_scalars(background_params(VARIANTS_PARAMS[ic])),
                                                                 g[pre + 'growth_k0'], of=of), rtol=1e-11)
            np.testing.assert_allclose(fo.pk_interpolator(of=of)(k, z), g[pre + 'pk_' + of], rtol=1e-10, err_msg=of)
        np.testing.assert_allclose(tr.transfer_kz(k[:4], z, grid=False), g[pre + 'transfer_delta_m'][np.arange(4), np.arange(4)], rtol=1e-11)
        np.testing.assert_allclose(fo.pk_interpolator(of='theta_m')(k, z), g[pre + 'pk_theta'], rtol=1e-10)
        np.testing.assert_allclose(fo.sigma8_m, g[pre + 'sigma8_m'], rtol=1e-10)
        np.testing.assert_allclose(fo.sigma8_z(z), g[pre + 'sigma8_z'], rtol=1e-10)
        np.testing.assert_allclose(eng._rsigma8, g[pre + 'rsigma8'], rtol=1e-10)
        np.testing.assert_allclose(cosmo.get_primordial().A_s, g[pre + 'A_s'], rtol=1e-10)
        np.testing.assert_allclose(cosmo.get_thermodynamics().rs_drag, g[pre + 'rs_drag_th'], rtol=1e-13)
        cross = fo.pk_interpolator(of=('delta_m', 'delta_cb'))(k, z)
        np.testing.assert_allclose(cross, np.sqrt(g[pre + 'pk_delta_m'] * g[pre + 'pk_delta_cb']), rtol=1e-9)
        with pytest.raises(cp.CosmologyError):
            tr.transfer_kz(k, z, of='delta_b')
