"""GPU: the sampler bindings against the reference's own adapters.  tests/golden/bindings.npz holds what the reference's
CosmoSIS module and Cobaya component wrote / returned when run on its analytic engine against the framework stand-ins of
oracle/framework_stubs.py (neither framework is in the image; oracle/gen_bindings_golden.py is the generating script); here this package's
adapters are driven through the same stand-ins with the same inputs."""
import os
import sys
import warnings

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import framework_stubs as stubs      # noqa: E402
from oracle.gen_bindings_golden import COSMOSIS_OPTIONS, COSMOSIS_PARAMS, COBAYA_PARAMS, COBAYA_REQUIREMENTS      # noqa: E402  (inputs only: no reference import)

RTOL = 1e-9


def test_cosmosis_module(golden):
    g = golden('bindings')
    names = stubs.install_cosmosis_stub()
    sys.modules.pop('cosmoprimo_amd.bindings.cosmosis', None)
    from cosmoprimo_amd.bindings import cosmosis as module
    warnings.simplefilter('ignore')
    config = module.setup(stubs.Options(COSMOSIS_OPTIONS))
    assert config['engine'] == 'eisenstein_hu' and config['nz'] == 9 and config['kmax'] == 50.
    block = stubs.DataBlock()
    for name, value in COSMOSIS_PARAMS.items():
        block[names.cosmological_parameters, name] = value
    block[names.cosmological_parameters, 'tcmb'] = 2.726
    assert module.execute(block, config) == 0
    checked = 0
    for key in sorted(g):
        if not key.startswith('cosmosis/'):
            continue
        _, section, name = key.split('/')
        if name == 'tcmb':
            continue
        got, ref = np.asarray(block[section, name], dtype='f8'), g[key]
        assert got.shape == ref.shape, key
        np.testing.assert_allclose(got, ref, rtol=RTOL, atol=0, err_msg=key)
        checked += 1
    assert checked == 27
    # a cosmology that cannot be computed is reported, not raised; CMB spectra are refused
    bad = stubs.DataBlock(block)
    bad[names.cosmological_parameters, 'h0'] = -1.
    assert module.execute(bad, config) == 1
    assert module.execute(stubs.DataBlock(block), dict(config, harmonic=True)) == 1
    assert module.cleanup(config) == 0


def test_cobaya_component(golden):
    g = golden('bindings')
    stubs.install_cobaya_stub()
    sys.modules.pop('cosmoprimo_amd.bindings.cobaya', None)
    from cosmoprimo_amd.bindings import cobaya as module
    warnings.simplefilter('ignore')
    theory = module.cosmoprimo(dict(engine='eisenstein_hu', extra_args={}, renames={}, output_params=['sigma8_m', 'Omega_m'], input_params=list(COBAYA_PARAMS)))
    theory.initialize()
    requirements = {k: (dict(v) if isinstance(v, dict) else v) for k, v in COBAYA_REQUIREMENTS.items()}
    requirements['Pk_grid'] = {'nonlinear': False, 'z': np.array([0., 0.5, 1.]), 'k_max': 2., 'vars_pairs': [('delta_tot', 'delta_tot')]}
    theory.must_provide(**requirements)
    theory.must_provide(Hubble={'z': np.array([0.3, 4.])})            # a second likelihood: pools of redshifts grow
    state = {'params': dict(COBAYA_PARAMS)}
    theory.calculate(state, want_derived=True, **COBAYA_PARAMS)
    z = np.asarray(COBAYA_REQUIREMENTS['Hubble']['z'])
    pick = theory.requests['Hubble'].pool.find_indices(z)
    np.testing.assert_allclose(state['Hubble'][pick], g['cobaya/Hubble'], rtol=RTOL)
    assert state['Hubble'].size == z.size + 1
    for name in ('angular_diameter_distance', 'comoving_radial_distance', 'angular_diameter_distance_2', 'sigma8_z', 'fsigma8'):
        np.testing.assert_allclose(state[name], g['cobaya/' + name], rtol=RTOL, err_msg=name)
    for kind in ('derived', 'derived_extra'):
        for name, value in state[kind].items():
            np.testing.assert_allclose(value, g['cobaya/%s/%s' % (kind, name)], rtol=RTOL, err_msg=name)
    assert set(state['derived']) == {'sigma8_m', 'Omega_m'} and set(state['derived_extra']) == {'rs_drag', 'Omega_m'}
    zs, radii, sigma = state[('sigma_R', 'delta_tot', 'delta_tot')]
    h = COBAYA_PARAMS['H0'] / 100.
    np.testing.assert_allclose(zs, g['cobaya/sigma_R.delta_tot.delta_tot/0'])
    np.testing.assert_allclose(radii * h, g['cobaya/sigma_R.delta_tot.delta_tot/1'], rtol=1e-14)      # the reference hands back R h (see the module's notes)
    np.testing.assert_allclose(sigma, g['cobaya/sigma_R.delta_tot.delta_tot/2'], rtol=RTOL)
    # P(k, z) grid: Cobaya's units (1/Mpc, Mpc^3), against the engine's own interpolator (the reference's component cannot run this product on
    # an analytic engine: oracle/gen_bindings_golden.py)
    k, zg, pk = state[('Pk_grid', False, 'delta_tot', 'delta_tot')]
    assert pk.shape == (zg.size, k.size) and k[0] == 1e-4 and np.isclose(k[-1], 2.)
    direct = np.asarray(theory.get_fourier().pk_interpolator(of='delta_m')(k / h, zg, grid=True)).T / h**3
    np.testing.assert_allclose(pk, direct, rtol=1e-12)
    assert np.all(np.diff(zg) < 0) and (pk[-1] > pk[0]).all()        # redshifts descending: the last row is z = 0
    # theta as the sampled parameter: h is solved for
    params = dict(COBAYA_PARAMS)
    params.pop('H0')
    params['theta_MC_100'] = float(module.parameter_of(theory.cosmo, 'theta_MC_100'))
    theory.calculate({'params': params}, want_derived=False, **params)
    assert abs(theory.cosmo.h - h) < 1e-5
    with pytest.raises(stubs.LoggedError):
        theory.must_provide(Cl={'tt': 2000})
    theory.close()
