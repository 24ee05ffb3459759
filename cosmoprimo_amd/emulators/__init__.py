"""
Batch driver under the reference's name (cosmoprimo/emulators/__init__.py:11-60): ``get_calculator(cosmo)`` turns a cosmology into a
function params -> dictionary of arrays, keyed '<section>.<quantity>' exactly as the reference's calculator ('background.comoving_radial_distance',
'thermodynamics.rs_drag', 'primordial.A_s', 'fourier.k', ...).  This is the caller that feeds batches of cosmologies to the device kernels
(training sets of emulators, samplers): every parameter may be an array of B values, and every returned array then carries B as leading axis
-- one launch of each kernel for the whole batch instead of the reference's one clone per sample.

Quantities follow the state dictionaries of the reference's emulated sections (cosmoprimo/emulators/emulated.py:218-223 background,
245-251 thermodynamics, 359-361 primordial, 581-600 fourier) on the same default grids (emulated.py:13-33).  The reference's calculator
silently drops 'fourier.pk.*' for its analytic engines (their ``pk_interpolator`` rejects the ``non_linear`` keyword and the error is
swallowed); here the pairs the analytic engines do provide ('delta_m', 'theta_m') are returned.  The neural-network emulator classes of that
module are out of scope (not on the hot path).
"""
import numpy as np

from ..cosmology import Cosmology, CosmologyError, _Sections
from ..interpolator import _host


class CalculatorComputationError(Exception):

    """Raised by the calculator when the cosmology could not be computed (reference emulators/tools/base.py)."""


def get_default_k_callable():
    """Default wavenumbers of 'fourier.k' [h/Mpc]: 422 values, denser towards high k (reference emulated.py:13-24)."""
    decades = [np.logspace(lo, lo + 1, num=num, endpoint=lo == 0) for lo, num in zip(range(-5, 1), range(20, 140, 20))]
    return np.concatenate([[1e-6]] + decades + [[1e2]])


def get_default_z_callable(key='fourier', non_linear=False):
    """Default redshifts: 256 values uniform in log(a) down to a = 1e-3 for the background, else 30 values up to z = 10 (emulated.py:27-33)."""
    if 'background' in key:
        return 1. / np.logspace(-3, 0., 256)[::-1] - 1.
    z = np.linspace(0., 10.**0.5, 30)**2
    return z[z < 2.] if non_linear else z


def _background_state(ba):
    z = get_default_z_callable('background')
    state = {'z': z}
    for name in ['rho_ncdm', 'p_ncdm', 'rho_fld', 'time', 'comoving_radial_distance']:
        state[name] = _host(getattr(ba, name)(z))
        if name.endswith('_ncdm') and ba._engine.batch_size:   # (N_ncdm, B, nz) -> batch axis first
            state[name] = np.moveaxis(state[name], 1, 0)
    return state


def _thermodynamics_state(th):
    state = {}
    for name in ['rs_drag', 'z_drag', 'rs_star', 'z_star', 'YHe']:
        try:
            state[name] = _host(getattr(th, name))
        except (AttributeError, CosmologyError):
            pass
    return state


def _primordial_state(pm):
    return {'A_s': _host(pm.A_s)}


def _fourier_state(fo):
    k, z = get_default_k_callable(), get_default_z_callable()
    state = {'k': k, 'z': z}
    for of in [('delta_m', 'delta_m'), ('delta_m', 'theta_m'), ('theta_m', 'theta_m')]:
        try:
            state['pk.{}.{}'.format(*of)] = _host(fo.pk_interpolator(of=of)(k, z))
        except (CosmologyError, NotImplementedError, ValueError, TypeError):   # a pair this engine does not provide
            pass
    return state


_states = {'background': _background_state, 'thermodynamics': _thermodynamics_state, 'primordial': _primordial_state, 'fourier': _fourier_state}


def get_calculator(cosmo, section=None):
    """
    Turn input cosmology into calculator:

    .. code-block:: python

        cosmo = Cosmology(engine='eisenstein_hu')
        calculator = get_calculator(cosmo)
        calculator(Omega_m=0.2)  # {'background.comoving_radial_distance': (256,) array, 'fourier.pk.delta_m.delta_m': (422, 30) array, ...}
        calculator(Omega_m=np.linspace(0.2, 0.4, 10000))  # same keys, arrays of shape (10000, 256), (10000, 422, 30), ...

    Anything that is not a :class:`Cosmology` is returned unchanged, as in the reference.
    """
    if not isinstance(cosmo, Cosmology):
        return cosmo
    if section is None:
        section = [name.lower() for name in _Sections if name.lower() in cosmo.engine._Sections]
    elif isinstance(section, str):
        section = [section]
    order = ['background', 'thermodynamics', 'primordial', 'perturbations', 'transfer', 'fourier', 'harmonic'][::-1]   # reference :28-32
    section_names = [name for name in order + [name for name in section if name not in order] if name in section]

    def calculator(**params):
        toret = {}
        try:
            clone = cosmo.clone(**params)
            for section_name in section_names:
                getstate = _states.get(section_name, None)
                if getstate is None:
                    continue
                for name, value in getstate(getattr(clone, 'get_{}'.format(section_name))()).items():
                    toret['{}.{}'.format(section_name, name)] = value
        except CosmologyError as exc:
            raise CalculatorComputationError from exc
        return toret

    return calculator
