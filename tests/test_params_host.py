"""Host-side parameter compilation (no GPU): massive-neutrino inputs against values produced by the reference in this container
(cosmology.py:960-1140): masses from Omega_ncdm / omega_ncdm (Newton), neutrino_hierarchy splitting, N_ur from N_eff."""
import numpy as np
import pytest


def test_ncdm_inputs():
    from cosmoprimo_amd.cosmology import _compile_params, CosmologyInputError
    np.testing.assert_allclose(_compile_params(dict(Omega_ncdm=[0.0014, 0.003], h=0.68))['m_ncdm'], [0.060294267293085214, 0.1292070888252256], rtol=1e-13)
    np.testing.assert_allclose(_compile_params(dict(Omega_ncdm=0.0014))['m_ncdm'], [0.0638935], rtol=1e-6)
    np.testing.assert_allclose(_compile_params(dict(omega_ncdm=0.00064))['m_ncdm'], [0.0596087], rtol=1e-6)
    assert _compile_params(dict(Omega_ncdm=0.))['m_ncdm'] == [0.]
    p = _compile_params(dict(m_ncdm=0.12, neutrino_hierarchy='normal'))
    np.testing.assert_allclose(p['m_ncdm'], [0.030108750535617665, 0.031311928379070764, 0.05857932108531181], rtol=1e-14)
    assert len(p['T_ncdm_over_cmb']) == 3
    np.testing.assert_allclose(p['N_ur'], 3.044 - 3 * 0.71611**4 * (4. / 11.)**(-4. / 3.), rtol=0, atol=1e-15)   # sum of three equal terms (cancellation)
    assert _compile_params(dict(m_ncdm=[0.06], N_ur=2.03))['N_ur'] == 2.03
    assert _compile_params({})['m_ncdm'] == [] and _compile_params({})['N_ur'] == 3.044
    with pytest.raises(CosmologyInputError):
        _compile_params(dict(m_ncdm=0.06, Omega_ncdm=0.001))
    with pytest.raises(TypeError):
        _compile_params(dict(m_ncdm=[0.06, 0.1], T_ncdm_over_cmb=[0.7]))
    with pytest.raises(CosmologyInputError):
        _compile_params(dict(w0_fld=-0.5, wa_fld=1.))
