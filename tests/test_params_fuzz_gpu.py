"""Parameter conventions drawn at random -- h or H0, Omega_x or omega_x, Omega_m or Omega_cdm, m_ncdm as a list or as a sum with a hierarchy, its own
temperature, N_eff or N_ur, T_cmb or Omega_g, curvature, (w0, wa), A_s or ln10^{10}A_s or logA or sigma8 -- through ``Cosmology(**params)``: the
compiled parameters for a fixed list of names, the masses, and what the Eisenstein-Hu engine makes of the amplitude (A_s, sigma8, rs_drag), against the
reference (tests/golden/params_fuzz.npz, `python -m oracle.gen_golden params_fuzz`; reference cosmology.py:1049-1260)."""
import numpy as np
import pytest

from oracle.gen_golden import params_fuzz_configs, params_fuzz_output, PARAMS_FUZZ_NAMES, PARAMS_FUZZ_N

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('i', range(PARAMS_FUZZ_N))
def test_random_conventions(golden, i):
    import torch
    assert torch.cuda.is_available()
    import cosmoprimo_amd as cp
    g = golden('params_fuzz')
    par = params_fuzz_configs()[i]
    values, masses = params_fuzz_output(cp, par)
    names = PARAMS_FUZZ_NAMES + ['A_s (engine)', 'sigma8_m', 'rs_drag']
    for name, got, ref in zip(names, values, g['values'][i]):
        rtol = 1e-9 if name in ('A_s (engine)', 'sigma8_m') else 1e-12
        np.testing.assert_allclose(got, ref, rtol=rtol, atol=1e-300, err_msg='%s of %s' % (name, par))
    np.testing.assert_allclose(masses, g['m_ncdm'][i], rtol=1e-12, equal_nan=True, err_msg=str(par))
