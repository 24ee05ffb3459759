"""Golden vectors of the reference's sampler bindings (bindings/cobaya/cosmoprimo.py, bindings/cosmosis/cosmoprimo_interface.py), generated in the
build container by RUNNING the reference's adapters on its analytic engine against stand-in framework objects (neither Cobaya nor CosmoSIS
is installed here; the stand-ins below are test infrastructure written for this purpose: the handful of calls the adapters make on a
DataBlock / on BoltzmannBase).  TEST INFRASTRUCTURE: writes tests/golden/bindings.npz and nothing else.

    python -m oracle.gen_bindings_golden
"""
import os
import sys
import types

import numpy as np

from ._refimport import import_reference
from .framework_stubs import install_cosmosis_stub, install_cobaya_stub, DataBlock, Options

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

COSMOSIS_OPTIONS = dict(zmin=0., zmax=2., nz=9, fourier=True, harmonic=False, engine='eisenstein_hu')
COSMOSIS_PARAMS = {'A_s': 2.1e-9, 'n_s': 0.965, 'h0': 0.68, 'ombh2': 0.0224, 'omch2': 0.12, 'omega_k': 0., 'tau': 0.054, 'w': -0.95, 'wa': 0.1}      # no 'mnu': the adapter's default, 0.06 eV (cosmoprimo_interface.py:44)

COBAYA_PARAMS = dict(H0=68., omega_b=0.0224, omega_cdm=0.12, A_s=2.1e-9, n_s=0.965, tau_reio=0.054)
COBAYA_Z = np.array([0., 0.3, 0.7, 1.1, 2.])
COBAYA_REQUIREMENTS = {
    'Hubble': {'z': COBAYA_Z}, 'angular_diameter_distance': {'z': COBAYA_Z}, 'comoving_radial_distance': {'z': COBAYA_Z},
    'angular_diameter_distance_2': {'z_pairs': [(0.1, 0.5), (0.3, 1.2), (0., 2.)]},
    'sigma8_z': {'z': COBAYA_Z[:3]}, 'fsigma8': {'z': COBAYA_Z[:3]},
    # ('Pk_grid' is left out: the reference's collector hands non_linear / k_max / z to pk_interpolator(), which its analytic engines reject --
    # TypeError at eisenstein_hu.py:328 -- so that product only runs with the Boltzmann codes, which are not in the image)
    'sigma_R': {'z': np.array([0., 1.]), 'R': np.array([4., 8., 12.]), 'k_max': 2., 'vars_pairs': [('delta_tot', 'delta_tot')]},
    'rs_drag': None, 'Omega_m': None,
}


def gen_cosmosis(out):
    names = install_cosmosis_stub()
    import_reference()
    from cosmoprimo.bindings.cosmosis import cosmoprimo_interface as ref
    config = ref.setup(Options(COSMOSIS_OPTIONS))
    block = DataBlock()
    for name, value in COSMOSIS_PARAMS.items():
        block[names.cosmological_parameters, name] = value
    block[names.cosmological_parameters, 'tcmb'] = 2.726
    assert ref.execute(block, config) == 0
    for (section, name), value in block.items():
        if (section, name) in [(names.cosmological_parameters, key) for key in COSMOSIS_PARAMS] or isinstance(value, str):
            continue
        out['cosmosis/%s/%s' % (section, name)] = np.asarray(value, dtype='f8')


def gen_cobaya(out):
    install_cobaya_stub()
    import_reference()
    import importlib
    ref_module = importlib.import_module('cosmoprimo.bindings.cobaya.cosmoprimo')
    theory = ref_module.cosmoprimo()
    theory.engine, theory.extra_args, theory.renames = 'eisenstein_hu', {}, {}
    theory.output_params, theory.input_params = ['sigma8_m', 'Omega_m'], list(COBAYA_PARAMS)
    theory.cosmoprimo_module = sys.modules['cosmoprimo']
    theory.derived_extra = []
    theory.must_provide(**{k: (dict(v) if isinstance(v, dict) else v) for k, v in COBAYA_REQUIREMENTS.items()})
    state = {'params': dict(COBAYA_PARAMS)}
    theory.calculate(state, want_derived=True, **COBAYA_PARAMS)
    for key, value in state.items():
        if key == 'params':
            continue
        tag = 'cobaya/' + ('.'.join(str(k) for k in key) if isinstance(key, tuple) else key)
        if isinstance(value, dict):
            for name, v in value.items():
                out['%s/%s' % (tag, name)] = np.asarray(v, dtype='f8')
        elif isinstance(value, tuple):
            for i, v in enumerate(value):
                out['%s/%d' % (tag, i)] = np.asarray(v, dtype='f8')
        else:
            out[tag] = np.asarray(value, dtype='f8')


def main():
    out = {}
    gen_cosmosis(out)
    gen_cobaya(out)
    path = os.path.join(ROOT, 'tests', 'golden', 'bindings.npz')
    np.savez_compressed(path, **out)
    for name in sorted(out):
        print(name, out[name].shape)
    print('wrote', path, os.path.getsize(path), 'bytes')


if __name__ == '__main__':
    main()
