"""GPU: the EH98 transfer function at the edges of its fast paths -- k = 0, negative, NaN and subnormal wavenumbers take the library's log (out
of line), arguments of the sine above 1e6 the library's sin -- against the oracle; everything else goes through the short log / sin / exp /
reciprocal forms, whose agreement with the operation-for-operation oracle is the subject of tests/test_cosmology_gpu.py."""
import warnings

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_transfer_edges():
    import cosmoprimo_amd as cp
    from oracle import power as opw
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        cosmo = cp.Cosmology(engine='eisenstein_hu')
        tr = cosmo.get_transfer()
        k = np.array([0., -1., np.nan, 1e-320, 1e-12, 1e-8, 1e-3, 1., 1e3, 9e3, 1.1e4, 1e5, 1e7, 1e9])
        got = np.asarray(tr.transfer_k(k))
        ba = cosmo.get_background()
        with np.errstate(all='ignore'):
            ref = opw.transfer_eh(k, cosmo['h'], opw.eh_scalars(cosmo['h'], cosmo['Omega_cdm'], cosmo['Omega_b'], cosmo['T_cmb'])).ravel()
    assert got[0] == 1. and np.isnan(got[1]) and np.isnan(got[2])
    np.testing.assert_allclose(got[3:], ref[3:], rtol=2e-11)
