"""Stand-ins for the two sampler frameworks the reference's bindings plug into (CosmoSIS's DataBlock / option section; Cobaya's BoltzmannBase,
pools of redshifts and logging), neither of which is installed in the build image.  TEST INFRASTRUCTURE: used by oracle/gen_bindings_golden.py
to run the reference's adapters, and by tests/test_bindings_*.py to run this package's adapters through the same calls.  Written from the
frameworks' documented behaviour for exactly the calls the adapters make; not part of the product."""
import logging
import sys
import types

import numpy as np

OPTION_SECTION = 'module_options'


class _Names(object):
    """cosmosis.datablock.names: attribute -> section name."""
    cosmological_parameters = 'cosmological_parameters'
    distances = 'distances'
    growth_parameters = 'growth_parameters'
    cmb_cl = 'cmb_cl'
    matter_power_lin = 'matter_power_lin'


class DataBlock(dict):
    """(section, name) -> value with the typed getters of cosmosis.datablock.DataBlock."""

    def _get(self, section, name, default):
        return dict.get(self, (section, name), default)

    def get_int(self, section, name, default=None):
        value = self._get(section, name, default)
        return value if value is None else int(value)

    def get_double(self, section, name, default=None):
        value = self._get(section, name, default)
        return value if value is None else float(value)

    def get_string(self, section, name, default=None):
        return self._get(section, name, default)

    def get_bool(self, section, name, default=None):
        value = self._get(section, name, default)
        return value if value is None else bool(value)

    def has_value(self, section, name):
        return (section, name) in self

    def put_grid(self, section, name_x, x, name_y, y, name_z, z):
        self[section, name_x], self[section, name_y], self[section, name_z] = np.asarray(x), np.asarray(y), np.asarray(z)

    def keys(self, section=None):
        return [key for key in dict.keys(self) if section is None or key[0] == section]


class Options(DataBlock):
    """The option section of a module's ini file."""

    def __init__(self, values):
        super().__init__({(OPTION_SECTION, name): value for name, value in values.items()})


def install_cosmosis_stub():
    pkg, datablock = types.ModuleType('cosmosis'), types.ModuleType('cosmosis.datablock')
    datablock.names, datablock.option_section = _Names, OPTION_SECTION
    pkg.datablock = datablock
    sys.modules['cosmosis'], sys.modules['cosmosis.datablock'] = pkg, datablock
    return _Names


# ---- Cobaya ----------------------------------------------------------------------------------------------------------------------------
def combine_1d(new, old=None):
    """cobaya.tools.combine_1d: sorted union without repetition."""
    new = np.atleast_1d(new)
    return np.sort(np.unique(np.concatenate([new, np.atleast_1d(old)]) if old is not None else new))


class Pool1D(object):
    """cobaya.tools.Pool1D: a growing sorted set of values; ``find_indices`` maps values back."""

    def __init__(self, values=()):
        self.values = np.empty(0)
        self.update(values)

    def update(self, values):
        self.values = combine_1d(values, self.values)

    def find_indices(self, values):
        idx = np.searchsorted(self.values, np.atleast_1d(values))
        assert np.allclose(self.values[idx], np.atleast_1d(values), rtol=1e-9, atol=0)
        return idx


class Pool2D(object):
    """cobaya.tools.Pool2D: a growing set of pairs, rows sorted lexicographically."""

    def __init__(self, values=()):
        self.values = np.empty((0, 2))
        self.update(values)

    def update(self, values):
        both = np.concatenate([self.values, np.atleast_2d(np.asarray(values, dtype='f8'))])
        self.values = np.unique(both, axis=0)

    def find_indices(self, values):
        values = np.atleast_2d(values)
        return np.array([int(np.flatnonzero(np.all(np.isclose(self.values, row, rtol=1e-9, atol=0), axis=1))[0]) for row in values])


class PoolND(object):
    pass


class LoggedError(Exception):
    def __init__(self, logger, msg='', *args):
        super().__init__(msg % args if args else msg)


class ComponentNotInstalledError(LoggedError):
    pass


class VersionCheckError(ValueError):
    pass


class BoltzmannBase(object):
    """The part of cobaya.theories.cosmo.BoltzmannBase a Boltzmann-code wrapper relies on: requirement bookkeeping, parameter renames, a logger."""
    renames = {}
    extra_args = None
    path = packages_path = None

    def __init__(self, info=None, **kwargs):
        self.log = logging.getLogger(type(self).__name__)
        self.extra_args = dict(self.extra_args or {})
        self.collectors, self._must_provide = {}, {}
        self.output_params, self.input_params = [], []
        self.current_state = {}
        for name, value in dict(info or {}, **kwargs).items():
            setattr(self, name, value)

    def initialize(self):
        self.collectors, self._must_provide = {}, {}

    def must_provide(self, **requirements):
        """Bookkeeping of cobaya.theories.cosmo.BoltzmannBase.must_provide: power-spectrum grids and sigma(R) are stored once per pair of fields under
        tuple keys -- ("Pk_grid", nonlinear, field, field) and ("sigma_R", field, field) -- with their redshifts merged; the rest as requested."""
        for key, value in requirements.items():
            if key in ('Pk_grid', 'Pk_interpolator'):
                value = dict(value)
                pairs = value.pop('vars_pairs', None) or [('delta_tot', 'delta_tot')]
                nonlinear = value.pop('nonlinear', True)
                for nl in (np.atleast_1d(nonlinear).tolist()):
                    for pair in pairs:
                        name = ('Pk_grid', bool(nl)) + tuple(sorted(pair))
                        known = self._must_provide.get(name, {})
                        self._must_provide[name] = dict(nonlinear=bool(nl), z=combine_1d(value['z'], known.get('z')),
                                                        k_max=max(value['k_max'], known.get('k_max', 0.)))
            elif key == 'sigma_R':
                value = dict(value)
                pairs = value.pop('vars_pairs', None) or [('delta_tot', 'delta_tot')]
                for pair in pairs:
                    name = ('sigma_R',) + tuple(sorted(pair))
                    known = self._must_provide.get(name, {})
                    self._must_provide[name] = dict(R=combine_1d(value['R'], known.get('R')), z=combine_1d(value['z'], known.get('z')),
                                                    k_max=max(value['k_max'], known.get('k_max', 0.)))
            else:
                self._must_provide[key] = value

    def translate_param(self, p):
        return self.renames.get(p, p)

    def check_no_repeated_input_extra(self):
        common = set(self.input_params).intersection(self.extra_args)
        if common:
            raise LoggedError(self.log, 'parameters both sampled and fixed: %r', common)

    def _cmb_unit_factor(self, units, T_cmb):
        return {'1': 1., 'muK2': T_cmb * 1e6, 'K2': T_cmb, 'FIRASmuK2': 2.7255e6, 'FIRASK2': 2.7255}[units]


def install_cobaya_stub():
    def module(name, **attrs):
        m = types.ModuleType(name)
        for key, value in attrs.items():
            setattr(m, key, value)
        sys.modules[name] = m
        return m

    nothing = lambda *args, **kwargs: None      # noqa: E731
    module('cobaya')
    module('cobaya.theories')
    module('cobaya.theories.cosmo', BoltzmannBase=BoltzmannBase)
    module('cobaya.log', LoggedError=LoggedError, get_logger=logging.getLogger)
    module('cobaya.install', download_github_release=nothing, pip_install=nothing, check_gcc_version=nothing)
    module('cobaya.component', ComponentNotInstalledError=ComponentNotInstalledError, load_external_module=nothing)
    module('cobaya.tools', Pool1D=Pool1D, Pool2D=Pool2D, PoolND=PoolND, combine_1d=combine_1d, get_compiled_import_path=nothing, VersionCheckError=VersionCheckError)
    return BoltzmannBase
