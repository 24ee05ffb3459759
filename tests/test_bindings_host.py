"""CPU: the host logic of the sampler bindings -- option / DataBlock translation of the CosmoSIS module, requirement bookkeeping of the Cobaya
component -- on the framework stand-ins of oracle/framework_stubs.py (no GPU: nothing is computed)."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import framework_stubs as stubs      # noqa: E402


def test_cosmosis_translation():
    names = stubs.install_cosmosis_stub()
    sys.modules.pop('cosmoprimo_amd.bindings.cosmosis', None)
    from cosmoprimo_amd.bindings import cosmosis as module
    config = module.setup(stubs.Options({'zmax': 1.5, 'fourier': True, 'cosmoprimo_k_pivot': 0.02}))
    assert config['zmin'] == 0. and config['zmax'] == 1.5 and config['nz'] == 150 and config['engine'] == 'eisenstein_hu' and config['cosmoprimo_k_pivot'] == 0.02
    block = stubs.DataBlock()
    for name, value in {'A_s': 2e-9, 'n_s': 0.96, 'h0': 0.7, 'ombh2': 0.022, 'omch2': 0.12, 'omega_k': 0.01, 'tau': 0.05, 'w': -0.9, 'nnu': 3.1,
                        'num_massive_neutrinos': 2, 'mnu': 0.05, 'sigma8': 0.8}.items():
        block[names.cosmological_parameters, name] = value
    block['reionization', 'delta_redshift'] = 0.4
    par = module.cosmology_parameters(block, config)
    assert par['H0'] == 70. and par['omega_b'] == 0.022 and par['omega_cdm'] == 0.12 and par['Omega_k'] == 0.01 and par['tau_reio'] == 0.05
    assert par['T_cmb'] == 2.726 and par['N_eff'] == 3.1 and par['w0_fld'] == -0.9 and 'wa_fld' not in par and par['reionization_width'] == 0.4
    assert par['m_ncdm'] == [0.05, 0.05] and par['neutrino_hierarchy'] is None
    assert par['k_pivot'] == 0.02 and par['sigma8'] == 0.8 and 'ombh2' not in par and 'mnu' not in par
    assert par['z_pk'].shape == (150,) and par['z_pk'][-1] == 1.5 and par['lensing'] is False
    block[names.cosmological_parameters, 'num_massive_neutrinos'] = 3
    block[names.cosmological_parameters, 'neutrino_hierarchy'] = 'normal'
    par = module.cosmology_parameters(block, config)
    assert par['m_ncdm'] == 0.05 and par['neutrino_hierarchy'] == 'normal'
    with pytest.raises(KeyError):
        module.cosmology_parameters(stubs.DataBlock(), config)


def test_cobaya_requirements():
    stubs.install_cobaya_stub()
    sys.modules.pop('cosmoprimo_amd.bindings.cobaya', None)
    from cosmoprimo_amd.bindings import cobaya as module
    theory = module.cosmoprimo(dict(extra_args={'kmax_pk': 0.5}, renames={'omegam': 'Omega_m', 'rdrag': 'rs_drag'}, output_params=['omegam'], input_params=['H0']))
    theory.initialize()
    theory.must_provide(Hubble={'z': [0.5, 0.1]}, angular_diameter_distance_2={'z_pairs': [(0.1, 0.5)]}, fsigma8={'z': [0.2]}, rdrag=None,
                        Pk_grid={'z': [0., 1.], 'k_max': 3., 'nonlinear': False, 'vars_pairs': [('delta_tot', 'delta_tot'), ('delta_nonu', 'delta_nonu')]})
    theory.must_provide(Hubble={'z': [0.3]}, angular_diameter_distance_2={'z_pairs': [(0.2, 0.5), (0.1, 0.5)]})
    assert np.array_equal(theory.requests['Hubble'].pool.values, [0.1, 0.3, 0.5])
    assert theory.requests['angular_diameter_distance_2'].pool.values.tolist() == [[0.1, 0.5], [0.2, 0.5]]
    assert theory.requests['fsigma8'].product.kwargs == {'of': 'theta_cb'}
    assert np.array_equal(theory.extra_args['z_pk'], [1., 0.2, 0.]) and theory.extra_args['kmax_pk'] == 3.
    assert theory.grids == {('Pk_grid', False, 'delta_tot', 'delta_tot'): ('delta_m', 'delta_m'), ('Pk_grid', False, 'delta_nonu', 'delta_nonu'): ('delta_cb', 'delta_cb')}
    assert theory.derived_extra == ['rs_drag']
    assert 'omegam' in theory.get_can_provide_params() and 'rdrag' in theory.get_can_provide_params()
    for bad in (dict(Cl={'tt': 100}), dict(nothing_known={'z': [0.]}), dict(Pk_grid={'z': [0.], 'k_max': 1., 'nonlinear': True})):
        with pytest.raises(stubs.LoggedError):
            theory.must_provide(**bad)


class _FakeCosmo(object):
    h, k_pivot, Omega0_m, Omega0_b, rs_drag = 0.5, 0.05, 0.3, 0.05, 100.

    def __getitem__(self, name):
        return {'theta_cosmomc': 0.0104, 'N_eff': 3.044, 'tau_reio': 0.}[name]


def test_cobaya_parameter_conventions():
    stubs.install_cobaya_stub()
    sys.modules.pop('cosmoprimo_amd.bindings.cobaya', None)
    from cosmoprimo_amd.bindings.cobaya import parameter_of
    cosmo = _FakeCosmo()
    assert parameter_of(cosmo, 'Omega_m') == 0.3 and parameter_of(cosmo, 'omega_b') == 0.05 * 0.25
    assert np.isclose(parameter_of(cosmo, 'theta_MC_100'), 1.04) and parameter_of(cosmo, 'k_pivot') == 0.025
    assert parameter_of(cosmo, 'N_eff') == 3.044 and parameter_of(cosmo, 'tau_reio') == 0. and parameter_of(cosmo, 'rs_drag') == 100.
