"""
Cosmology container, engine registry and the analytic engines on MI355X, behind the reference's API
(cosmoprimo/cosmology.py: ``Cosmology``, ``BaseEngine`` + metaclass registry :459-503, ``get_engine`` :574-633, section
getters :636-705, ``BaseBackground`` :1627-1933, ``DefaultBackground`` :1954-2093; engines eisenstein_hu.py,
eisenstein_hu_nowiggle.py, bbks.py).

Scope: the parameters the analytic engines and the background use (h / H0, Omega_cdm / omega_cdm /
Omega_m / omega_m, Omega_b / omega_b, Omega_k, sigma8 / A_s / logA, n_s, alpha_s, beta_s, k_pivot, T_cmb, N_eff / N_ur,
w0_fld, wa_fld, tau_reio ..., massive neutrinos through m_ncdm / Omega_ncdm / neutrino_hierarchy), the calculation parameters of the
Boltzmann engines carried but unread, ``clone`` in both bases, ``solve``, json / npy persistence and the section shortcuts.

Extension for the GPU: every numerical parameter may be an array (or torch tensor) of shape (B,): a *batch* of cosmologies.
All section methods then return results with a leading (B,) axis.
"""
import sys

import numpy as np

from . import _lib, background as bgmod, power as pwmod, utils
from . import _device as dv
from .interpolator import PowerSpectrumInterpolator1D, PowerSpectrumInterpolator2D, _host, _finish

TCMB, NEFF, TNCDM_OVER_CMB = 2.7255, 3.044, 0.71611   # cosmoprimo/constants.py


class CosmologyError(Exception):
    """Generic exception raised when an error related to cosmology arises (reference cosmology.py:38)."""


class CosmologyInputError(CosmologyError):
    """Exception raised when error in input parameters."""


class CosmologyComputationError(CosmologyError):
    """Exception raised when error in cosmology computation."""


_default_cosmological_parameters = dict(h=0.7, Omega_cdm=0.25, Omega_b=0.05, Omega_k=0., sigma8=0.8, k_pivot=0.05, n_s=0.96, alpha_s=0., beta_s=0.,
                                        r=0., n_t='scc', alpha_t='scc', T_cmb=TCMB, m_ncdm=None, neutrino_hierarchy=None, T_ncdm_over_cmb=TNCDM_OVER_CMB,
                                        N_eff=NEFF, tau_reio=0.06, reionization_width=0.5, A_L=1.0, w0_fld=-1., wa_fld=0., cs2_fld=1.)
# accepted and carried for the Boltzmann engines of the reference; the analytic engines of this package do not read them (cosmology.py:734)
_default_calculation_parameters = dict(non_linear='', modes='s', lensing=False, z_pk=None, kmax_pk=10., ellmax_cl=2500, YHe='BBN', use_ppf=True)
_conflict_parameters_no_alias = [('h', 'H0'), ('T_cmb', 'Omega_g', 'omega_g'), ('Omega_b', 'omega_b'),
                                 ('Omega_cdm', 'omega_cdm', 'Omega_c', 'omega_c', 'Omega_m', 'omega_m'), ('Omega_k', 'omega_k'),
                                 ('N_ur', 'Omega_ur', 'omega_ur', 'N_eff'), ('m_ncdm', 'Omega_ncdm', 'omega_ncdm'), ('A_s', 'logA', 'sigma8'),
                                 ('tau_reio', 'z_reio')]
_alias_parameters = {'omega_b': ('ombh2',), 'omega_cdm': ('omch2',), 'Omega_k': ('omk', 'Omega0_k'), 'm_ncdm': ('mnu',), 'N_eff': ('nnu',), 'n_s': ('ns',),
                     'alpha_s': ('nrun',), 'beta_s': ('nrunrun',), 'tau_reio': ('tau',), 'Omega_m': ('Omega0_m',), 'Omega_cdm': ('Omega0_cdm', 'Omega_c'),
                     'Omega_b': ('Omega0_b',), 'Omega_ur': ('Omega0_ur',), 'Omega_ncdm': ('Omega0_ncdm',), 'Omega_fld': ('Omega0_fld',),
                     'T_cmb': ('T0_cmb',), 'Omega_g': ('Omega0_g',), 'logA': ('ln10^10A_s', 'ln10^{10}A_s', 'ln_A_s_1e10'), 'w0_fld': ('w',),
                     'wa_fld': ('wa',)}


def _all_conflicts():
    """Groups of mutually exclusive input names: the physical groups extended by the aliases of their members, plus one group per remaining
    aliased name (same rule as reference cosmology.py:1543-1558, so that e.g. ``tau`` and ``tau_reio`` together are refused)."""
    groups = []
    for group in _conflict_parameters_no_alias:
        group = list(group)
        for name in list(group):
            group += [alias for alias in _alias_parameters.get(name, ()) if alias not in group]
        groups.append(tuple(group))
    for name, aliases in _alias_parameters.items():
        if not any(name in group for group in _conflict_parameters_no_alias):
            groups.append((name,) + tuple(aliases))
    return groups


_conflict_parameters = _all_conflicts()


def find_conflicts(name, conflicts=tuple()):
    """The group of ``conflicts`` that holds ``name`` (itself included), () if none -- and none for the default, empty, ``conflicts``, as the
    reference (cosmology.py:1606-1624): its callers pass ``conflicts=cls._conflict_parameters``."""
    for group in conflicts:
        if name in group:
            return group
    return ()


def check_params(args, **kwargs):
    """Raise :class:`CosmologyInputError` if two names of ``args`` exclude each other; ``kwargs`` (``conflicts=``) go to :func:`find_conflicts`
    (reference cosmology.py:1592-1603)."""
    for name in args:
        found = [other for other in find_conflicts(name, **kwargs) if other != name and other in args]
        if found:
            raise CosmologyInputError('Conflicting parameters are given: {}'.format([name] + found))


def merge_params(args, moreargs, **kwargs):
    """``moreargs`` into ``args`` (in place): a new name first removes everything it excludes; ``kwargs`` (``conflicts=``) go to
    :func:`find_conflicts` (reference cosmology.py:1561-1589)."""
    for name in moreargs:
        for other in find_conflicts(name, **kwargs):
            args.pop(other, None)
    args.update(moreargs)
    return args


def is_sequence(item):
    """Tuple or list? (reference cosmology.py:53-54)"""
    return isinstance(item, (tuple, list))


def _is_array(v):
    return dv.is_torch(v) or np.ndim(v) > 0


class class_or_instancemethod(classmethod):

    """A method bound to the instance when called on one, to the class otherwise (reference cosmology.py:16-19)."""

    def __get__(self, instance, type_):
        return (super().__get__ if instance is None else self.__func__.__get__)(instance, type_)


_class_or_instancemethod = class_or_instancemethod


def _deepeq(a, b):
    """Deep equality of parameter containers (dicts, sequences, arrays, device tensors, scalars)."""
    if type(a) is not type(b):
        return False
    if isinstance(a, dict):
        return a.keys() == b.keys() and all(_deepeq(a[name], b[name]) for name in a)
    if isinstance(a, (tuple, list)):
        return len(a) == len(b) and all(_deepeq(x, y) for x, y in zip(a, b))
    if dv.is_torch(a):
        return a.shape == b.shape and bool((a == b).all())
    if isinstance(a, np.ndarray):
        return a.shape == b.shape and bool(np.all(a == b))
    return a == b


def _to_json(obj):
    """State dictionary -> JSON-serialisable (arrays and device tensors as tagged lists)."""
    if isinstance(obj, dict):
        return {name: _to_json(value) for name, value in obj.items()}
    if isinstance(obj, (list, tuple)):
        return [_to_json(value) for value in obj]
    if dv.is_torch(obj):
        obj = dv.to_host(obj)
    if isinstance(obj, np.ndarray):
        return {'__array__': obj.tolist(), 'dtype': str(obj.dtype)}
    if isinstance(obj, np.generic):
        return obj.item()
    return obj


def _from_json(obj):
    if isinstance(obj, dict):
        if '__array__' in obj:
            return np.array(obj['__array__'], dtype=obj['dtype'])
        return {name: _from_json(value) for name, value in obj.items()}
    if isinstance(obj, list):
        return [_from_json(value) for value in obj]
    return obj


def _ncdm_momenta_z0(T_eff, m, out='rho'):
    """Phase-space integral of one massive species today (reference _compute_ncdm_momenta, cosmology.py:74-137, 100-point
    Gauss-Laguerre), in 1e10 Msun / Mpc^3: host-side parameter derivation (Omega_ncdm, Omega_m -> Omega_cdm), floats or (B,) arrays."""
    T_eff, m = np.asarray(_host(T_eff), dtype='f8'), np.asarray(_host(m), dtype='f8')
    over_T = 1.602176634e-19 / (1.380649e-23 * T_eff)
    m2 = ((m * over_T)**2)[..., None]
    ti, wi = np.polynomial.laguerre.laggauss(100)
    if out == 'rho':
        f = ti**2 * np.sqrt(ti**2 + m2) / (1. + np.exp(-ti))
    elif out == 'drhodm':      # derivative of the density w.r.t. the mass (per eV), for the Newton solve of Omega_ncdm -> m_ncdm
        f = (m * over_T**2)[..., None] * ti**2 / np.sqrt(ti**2 + m2) / (1. + np.exp(-ti))
    else:
        f = 1. / 3. * ti**4 / np.sqrt(ti**2 + m2) / (1. + np.exp(-ti))
    c, sb, _ = bgmod_constants()
    mpc, msun = 1e6 * 3.085677581491367e16, 1.98847 * 1e30
    return 7. / 8. * 4 / c**3 * sb * T_eff**4 * np.sum(f * wi, axis=-1) / (7. * np.pi**4 / 120.) / (1e10 * msun) * mpc**3


def compute_ncdm_momenta(T_eff, m, z, method='laguerre', epsabs=1e-7, epsrel=1e-7, out='rho'):
    """Density ('rho'), its derivative w.r.t. the mass ('drhodm', per eV) or pressure ('p') of one massive species of temperature ``T_eff`` today [K]
    and mass ``m`` [eV] at redshift ``z``, in 1e10 Msun / Mpc^3, by the reference's name and arguments (``_compute_ncdm_momenta``,
    cosmology.py:74-137): the integral depends on the temperature at z only, T_eff (1 + z).  ``method='laguerre'``: 100-point Gauss-Laguerre (what
    the background tables of this package are built from, on the device: ``cp_ncdm_tables``); ``method='quad'``: ``scipy.integrate.quad`` on
    (0, 100) with ``epsabs`` / ``epsrel``, redshift by redshift on the host, as the reference."""
    if out not in ('rho', 'drhodm', 'p'):
        raise ValueError('Cannot compute ncdm momenta {}; choices are ["rho", "drhodm", "p"]'.format(out))
    z = np.asarray(_host(z), dtype='f8')
    T = np.asarray(_host(T_eff), dtype='f8') * (1. + z)
    if method != 'quad':
        return _ncdm_momenta_z0(T, m, out=out)
    from scipy import integrate
    m = float(np.asarray(_host(m), dtype='f8'))
    over_T = 1.602176634e-19 / (1.380649e-23 * T.ravel())

    def integrand(q, m_over_T2, m2_over_T2):
        if out == 'rho':
            return q**2 * np.sqrt(q**2 + m2_over_T2) / (1. + np.exp(q))
        if out == 'drhodm':
            return m_over_T2 * q**2 / np.sqrt(q**2 + m2_over_T2) / (1. + np.exp(q))
        return 1. / 3. * q**4 / np.sqrt(q**2 + m2_over_T2) / (1. + np.exp(q))

    toret = np.array([integrate.quad(integrand, 0., 100., args=(m * ot**2, (m * ot)**2), epsabs=epsabs, epsrel=epsrel)[0] for ot in over_T])
    c, sb, _ = bgmod_constants()
    mpc, msun = 1e6 * 3.085677581491367e16, 1.98847 * 1e30
    return (7. / 8. * 4 / c**3 * sb * T.ravel()**4 * toret / (7. * np.pi**4 / 120.) / (1e10 * msun) * mpc**3).reshape(z.shape)


def get_default_z_interp(name):
    """The redshift knots the reference's background tabulates ``name`` on (cosmology.py:1940-1952): 'rho_ncdm' / 'p_ncdm' (119), 'time' / 'age' (400),
    'comoving_radial_distance' (119) -- the grids the kernels of ``cp_background.hip`` are built on, read back from the library."""
    if name in ('rho_ncdm', 'p_ncdm'):
        knots = np.empty(_lib.NCDM_NKNOTS)
        _lib.check(_lib.load().cp_ncdm_knots(_lib.as_double_p(knots), knots.size))
        return knots
    if name in ('time', 'age', 'comoving_radial_distance'):
        knots = np.empty(400 if name in ('time', 'age') else 119)
        _lib.check(_lib.load().cp_background_knots(_lib.as_double_p(knots), knots.size))
        return knots
    raise ValueError('No default z interpolation grid for {}'.format(name))


def _split_neutrino_masses(sum_ncdm, hierarchy):
    """Three masses adding up to ``sum_ncdm`` [eV] under the 'normal', 'inverted' or 'degenerate' hierarchy, by Newton's method on the
    lightest mass with the squared-mass splittings of arXiv:1907.12598 (reference cosmology.py:1047-1104)."""
    if sum_ncdm < 0.:
        raise CosmologyInputError('Parameter {} should be positive, found {}'.format('m_ncdm', sum_ncdm))
    deltam21sq = 7.39e-5

    def newton(m, d21, d31):
        m = list(m)
        check = sum(m)
        for _ in range(1000):
            if not abs(sum_ncdm - check) > 1e-15:
                break
            m[0] = m[0] + (sum_ncdm - check) / (1. + m[0] / m[1] + m[0] / m[2])
            m[1] = np.sqrt(m[0]**2 + d21)
            m[2] = np.sqrt(m[0]**2 + d31)
            check = sum(m)
        return [float(x) for x in m]

    if hierarchy == 'normal':
        deltam31sq = 2.525e-3
        if sum_ncdm**2 < deltam21sq + deltam31sq:
            raise CosmologyInputError('the normal neutrino hierarchy needs a total mass above sqrt(dm21^2 + dm31^2) ~ 0.0592 eV, got {:.4f} eV'.format(sum_ncdm))
        return newton([0., deltam21sq, deltam31sq], deltam21sq, deltam31sq)
    if hierarchy == 'inverted':
        deltam32sq = -2.512e-3
        deltam31sq = deltam32sq + deltam21sq
        if sum_ncdm**2 < -deltam31sq - deltam32sq:
            raise CosmologyInputError('the inverted neutrino hierarchy needs a total mass above sqrt(-dm31^2 - dm32^2) ~ 0.0978 eV, got {:.4f} eV'.format(sum_ncdm))
        return newton([np.sqrt(-deltam31sq), np.sqrt(-deltam32sq), 1e-5], deltam21sq, deltam31sq)
    if hierarchy == 'degenerate':
        return [sum_ncdm / 3.] * 3
    raise CosmologyInputError('Unkown neutrino mass type {}'.format(hierarchy))


def _compile_params(args):
    """Input parameters -> the canonical set (a reduced restatement of reference Cosmology._compile_params, cosmology.py:874-1217)."""
    check_params(args, conflicts=_conflict_parameters)
    params = {}
    for name, value in args.items():   # aliases
        for canon, aliases in _alias_parameters.items():
            if name in aliases:
                name = canon
        params[name] = value
    out = merge_params(dict(_default_cosmological_parameters, **_default_calculation_parameters), params, conflicts=_conflict_parameters)   # a given name removes the defaults it excludes
    hierarchy = out.pop('neutrino_hierarchy', None)
    if 'omega_ncdm' in out:
        out['Omega_ncdm'] = np.asarray(out.pop('omega_ncdm'), dtype='f8') / np.asarray(_host(out.get('h', out.get('H0', 70.) / 100.)), dtype='f8')**2
    if 'Omega_ncdm' in out:   # densities instead of masses: Newton's method on every mass (reference cosmology.py:975-1014); scalar cosmologies
        if out.get('m_ncdm', None) is not None:
            raise CosmologyInputError('Conflicting parameters: m_ncdm and Omega_ncdm')
        Omega_ncdm = out.pop('Omega_ncdm')
        Omega_ncdm = [] if Omega_ncdm is None else list(np.atleast_1d(np.asarray(Omega_ncdm, dtype='f8')))
        T_in = out.get('T_ncdm_over_cmb', None)
        T_list = [TNCDM_OVER_CMB if T_in is None else t for t in (np.broadcast_to(np.asarray(TNCDM_OVER_CMB if T_in is None else T_in, dtype='f8'), (len(Omega_ncdm),)))]
        hh = float(_host(out['H0'] / 100. if 'H0' in out else out['h']))
        rck = bgmod_constants()[2] / (1e10 * 1.98847 * 1e30) * (1e6 * 3.085677581491367e16)**3
        masses = []
        for Om, t in zip(Omega_ncdm, T_list):
            omega, T_eff = Om * hh**2, float(_host(out['T_cmb'])) * t
            m = omega * 93.14
            if Om != 0.:
                check = _ncdm_momenta_z0(T_eff, m, 'rho') / rck
                for _ in range(1000):
                    if not abs(omega - check) > 1e-15:
                        break
                    m = m + (omega - check) / (_ncdm_momenta_z0(T_eff, m, 'drhodm') / rck)
                    check = _ncdm_momenta_z0(T_eff, m, 'rho') / rck
            masses.append(float(m))
        out['m_ncdm'] = masses
    # massive neutrinos (reference cosmology.py:960-969, 1113-1140): one entry per species, each a float or a (B,) array
    m_ncdm = out.pop('m_ncdm', None)
    if m_ncdm is None:
        m_ncdm = []
    elif isinstance(m_ncdm, (list, tuple)):
        m_ncdm = list(m_ncdm)           # entries: a float, or a (B,) array for a batch of cosmologies
    else:
        m_ncdm = list(np.atleast_1d(_host(m_ncdm)))   # a scalar or a 1D array: the masses of the species
    m_ncdm = [m if _is_array(m) else float(m) for m in m_ncdm]
    T_over = out.pop('T_ncdm_over_cmb', None)
    if hierarchy is not None:   # the sum of masses split into three species (reference cosmology.py:1030-1106)
        single = np.ndim(params.get('m_ncdm', None)) == 0 and params.get('m_ncdm', None) is not None
        if not single or len(m_ncdm) != 1 or _is_array(m_ncdm[0]):
            raise CosmologyInputError('neutrino_hierarchy {} cannot be passed with a list for m_ncdm, only with a sum.'.format(hierarchy))
        m_ncdm = _split_neutrino_masses(m_ncdm[0], hierarchy)
        if T_over is not None and np.ndim(T_over) > 0:
            T_over = [list(T_over)[0]] * 3
    if T_over is None:
        T_over = TNCDM_OVER_CMB
    if np.ndim(T_over) == 0:
        T_over = [T_over] * len(m_ncdm)
    T_over = [float(t) for t in T_over]
    if len(T_over) != len(m_ncdm):
        raise TypeError('T_ncdm_over_cmb and m_ncdm must be of same length, found {:d} != {:d}'.format(len(T_over), len(m_ncdm)))
    if 'N_ncdm' in out and out.pop('N_ncdm') not in (None, len(m_ncdm)):
        raise ValueError('provided N_ncdm does not match len(m_ncdm) = {:d}'.format(len(m_ncdm)))
    out['m_ncdm'], out['T_ncdm_over_cmb'] = m_ncdm, T_over
    if 'H0' in out:
        out['h'] = out.pop('H0') / 100.
    h = out['h']
    for name in ['b', 'cdm', 'c', 'm', 'k', 'g', 'ur']:
        if 'omega_' + name in out:
            out['Omega_' + name] = out.pop('omega_' + name) / h**2
    if 'Omega_c' in out:
        out['Omega_cdm'] = out.pop('Omega_c')
    if 'Omega_g' in out:     # photon density -> temperature (cosmology.py:952-953)
        c, sb, rck = bgmod_constants()
        out['T_cmb'] = (out.pop('Omega_g') * h**2 * rck / (4. / c**3 * sb))**0.25
    if 'Omega_ur' in out:    # density of the massless species -> their number (cosmology.py:1110-1113)
        c, sb, rck = bgmod_constants()
        out['N_ur'] = out.pop('Omega_ur') / (7. / 8. * 4. / c**3 * sb * (out['T_cmb'] * (4. / 11.)**(1. / 3.))**4 / (h**2 * rck))
        out.pop('N_eff', None)
    if 'Omega_m' in out:   # Omega_cdm = Omega_m - Omega_b - non-relativistic part of the massive neutrinos, cosmology.py:1163-1165
        rck = bgmod_constants()[2] / (1e10 * 1.98847 * 1e30) * (1e6 * 3.085677581491367e16)**3     # rho_crit_over_Msunph_per_Mpcph3
        nonrel = sum((_ncdm_momenta_z0(_host(out['T_cmb']) * t, m, 'rho') - 3 * _ncdm_momenta_z0(_host(out['T_cmb']) * t, m, 'p')) for t, m in zip(T_over, m_ncdm))
        nonrel = nonrel / np.asarray(_host(h), dtype='f8')**2 / rck if m_ncdm else 0.
        if np.ndim(nonrel) == 0:
            nonrel = float(nonrel)
        elif dv.is_torch(out['Omega_m']) or dv.is_torch(out['Omega_b']):
            nonrel = dv.torch().as_tensor(nonrel, device=(out['Omega_m'] if dv.is_torch(out['Omega_m']) else out['Omega_b']).device)
        out['Omega_cdm'] = out.pop('Omega_m') - out['Omega_b'] - nonrel
    if 'N_ur' not in out:  # N_ur = N_eff - sum over massive species (T_ncdm / T_cmb)^4 (4/11)^(-4/3), cosmology.py:1123-1127
        out['N_ur'] = out.pop('N_eff') - sum(t**4 * (4. / 11.)**(-4. / 3.) for t in T_over)
    else:
        out.pop('N_eff', None)
    if 'logA' in out:
        out['A_s'] = np.exp(out.pop('logA')) * 1e-10 if not dv.is_torch(out['logA']) else dv.torch().exp(out.pop('logA')) * 1e-10
    w0, wa = out['w0_fld'], out['wa_fld']
    if not _is_array(w0) and not _is_array(wa) and w0 + wa >= 1. / 3.:   # cosmology.py:1171-1177
        raise CosmologyInputError('w(a -> 0) = w0_fld + wa_fld > 1 / 3 (found {:.2f}), violates radiation domination at early time'.format(w0 + wa))
    # calculation parameters (cosmology.py:1150-1161): carried, not read by the analytic engines
    if out.get('z_pk', None) is None:
        out['z_pk'] = np.linspace(0., 10.**0.5, 30)**2     # interpolator.get_default_z_callable of the reference
    if out.get('modes', None) is None:
        out['modes'] = ['s']
    for name in ['modes', 'z_pk']:
        if np.ndim(out[name]) == 0:
            out[name] = [out[name]]
    out['z_pk'] = np.sort(np.asarray(out['z_pk'], dtype='f8'))
    if 0. not in out['z_pk']:
        out['z_pk'] = np.insert(out['z_pk'], 0, 0.)
    out['use_ppf'] = bool(out.get('use_ppf', True))
    for name in ['Omega_cdm', 'Omega_b', 'T_cmb', 'h', 'A_s', 'sigma8', 'm_ncdm', 'T_ncdm_over_cmb']:   # cosmology.py:1182-1192
        values = out.get(name, None)
        for value in (values if isinstance(values, (list, tuple)) else [values]):
            if value is not None and not dv.is_torch(value) and bool(np.any(np.asarray(value, dtype='f8') < 0.)):   # device batches are not read back
                raise CosmologyInputError('Parameter {} should be positive, found {}'.format(name, value))
    for name, allowed in [('YHe', ('BBN',)), ('n_t', ('SCC',)), ('alpha_t', ('SCC',))]:   # a float or the named rule (cosmology.py:1194-1209)
        value = out.get(name, None)
        value = allowed[0] if value is None else value
        if isinstance(value, str):
            if value.upper() not in allowed:
                raise CosmologyInputError('Parameter {} should be either a float or one of {}'.format(name, allowed))
            value = value.upper()
        out[name] = value
    r, n_s = out['r'], out['n_s']
    # no tensors (r = 0, the default) and a batch of tilts: both rules give zero for every cosmology -- the number, not six kernels on (B,) arrays
    no_tensors = not _is_array(r) and r == 0. and _is_array(n_s)
    if isinstance(out['n_t'], str):        # single-field slow-roll consistency (cosmology.py:1212-1215)
        out['n_t'] = 0. if no_tensors else - r / 8.0 * (2.0 - n_s - r / 8.0)
    if isinstance(out['alpha_t'], str):
        out['alpha_t'] = 0. if no_tensors else r / 8.0 * (r / 8.0 + n_s - 1)
    return out


_missing = object()
_DEVICE_DERIVED = frozenset(_lib.DERIVED_VALUES)


class BaseCosmoParams(dv.Copyable):

    """Parameter access shared by :class:`Cosmology` and engines (reference BaseCosmoParams, cosmology.py:231-457)."""

    @classmethod
    def get_default_params(cls, of=None, include_conflicts=True):
        """Default input parameters ``of`` 'cosmology', 'calculation' or both (None); with ``include_conflicts`` every accepted name appears,
        the excluded alternatives carrying the value of the default they exclude (reference cosmology.py:250-288)."""
        if of is None:
            toret = cls.get_default_params(of='cosmology', include_conflicts=include_conflicts)
            toret.update(cls.get_default_params(of='calculation', include_conflicts=include_conflicts))
            return toret
        if of not in ('cosmology', 'calculation'):
            raise CosmologyInputError('No default parameters for {}'.format(of))
        toret = dict(_default_cosmological_parameters if of == 'cosmology' else _default_calculation_parameters)
        if include_conflicts:
            for name in list(toret):
                for other in find_conflicts(name, conflicts=_conflict_parameters):
                    toret[other] = toret[name]
        return toret

    @classmethod
    def get_default_parameters(cls, **kwargs):
        import warnings
        warnings.warn('get_default_parameters is deprecated, use get_default_params')
        return cls.get_default_params(**kwargs)

    def get_params(self, of='base'):
        """Parameters ``of`` 'base' (compiled), 'cosmology', 'calculation', 'derived', 'extra' (engine) or 'all' (reference cosmology.py:290-320)."""
        if of == 'derived':
            return dict(getattr(self, '_derived', {}))
        if of == 'extra':
            return dict(getattr(self, '_extra_params', {}))
        toret = dict(self._params)
        if of == 'base':
            return toret
        if of in ('cosmology', 'calculation'):
            return {name: toret.get(name, value) for name, value in self.get_default_params(of=of).items()}
        if of == 'all':
            toret.update(self.get_params(of='derived'))
            toret.update(self.get_params(of='extra'))
            return toret
        raise CosmologyInputError('No parameters for {}'.format(of))

    def __eq__(self, other):
        """Same parameters (and engine parameters)? (reference cosmology.py:454-456)"""
        return type(other) == type(self) and _deepeq(other._params, self._params) and _deepeq(getattr(other, '_extra_params', {}), getattr(self, '_extra_params', {}))

    __hash__ = object.__hash__

    def __getitem__(self, name):
        return self.get(name)

    def __contains__(self, name):
        try:
            self.get(name)
            return True
        except CosmologyError:
            return False

    def get(self, *args, **kwargs):
        """Return an input (or easily derived) parameter (reference cosmology.py:331-415)."""
        if len(args) == 1:
            name, has_default, default = args[0], 'default' in kwargs, kwargs.get('default', None)
        else:
            (name, default), has_default = args, True
        params = self._params
        if name in params:
            return params[name]
        # derived values are kept: parameters do not change after construction, and sections read dozens of them (some cost a device read-back)
        memo = self.__dict__.setdefault('_derived_memo', {})
        if name in memo:
            return memo[name]
        if name in _DEVICE_DERIVED:      # a batch on the device: every derived value from ONE kernel, at the first that is asked for
            table = self._device_derived()
            if table is not None:
                memo.update(table)
                return memo[name]
        found = self._derive(name)
        if found is not _missing:
            memo[name] = found
            return found
        if has_default:
            return default
        raise CosmologyError('Parameter {} not found.'.format(name))

    def _h2(self):
        """h^2, kept with the derived values: half a dozen derivations use it, and for a batch of cosmologies every square is a device kernel."""
        memo = self.__dict__.setdefault('_derived_memo', {})
        if '_h2' not in memo:
            table = self._device_derived()
            if table is not None:
                memo.update(table)
            else:
                memo['_h2'] = self._params['h']**2
        return memo['_h2']

    def _device_derived(self):
        """{name: (B,) device tensor} of the derived parameters of a batch of cosmologies kept on the device (``cp_derived_parameters``: one
        launch for all of them; elementwise framework arithmetic took two to five launches for each), None for anything else -- host
        parameters, massive species (their densities are derived on the host)."""
        params = self._params
        if params.get('m_ncdm'):
            return None
        tensors = [params[name] for name in _lib.BG_PARAMS if dv.is_torch(params[name])]
        if not tensors or not all(t.is_cuda and t.ndim == 1 and t.dtype == tensors[0].dtype for t in tensors):
            return None
        torch, device = dv.torch(), tensors[0].device
        if tensors[0].dtype != torch.float64:
            return None
        carr, n, keep = dv.pack_params(_lib.BG_PARAMS, params, bgmod.DEFAULTS, device)
        if n is None:
            return None
        out = torch.empty((len(_lib.DERIVED_VALUES), n), dtype=torch.float64, device=device)
        _lib.check(_lib.load().cp_derived_parameters(n, dv.as_void_p(carr), out.data_ptr(), device.index, dv.stream_of(device)))
        return {name: out[i] for i, name in enumerate(_lib.DERIVED_VALUES)}

    def _derive(self, name):
        """``name`` from the compiled parameters, ``_missing`` if it is not a derived parameter."""
        params = self._params
        c, sb, rck = bgmod_constants()
        if name.startswith('omega'):
            return self.get('O' + name[1:]) * self._h2()
        if name == 'H0':
            return params['h'] * 100
        if name in ['logA', 'ln10^{10}A_s', 'ln10^10A_s', 'ln_A_s_1e10'] and 'A_s' in params:
            return np.log(1e10 * params['A_s'])
        if name == 'Omega_g':
            return params['T_cmb']**4 * 4. / c**3 * sb / (self._h2() * rck)
        if name == 'T_ur':
            return params['T_cmb'] * (4. / 11.)**(1. / 3.)
        if name == 'Omega_ur':
            return params['N_ur'] * 7. / 8. * self.get('T_ur')**4 * 4. / c**3 * sb / (self._h2() * rck)
        if name == 'Omega_r':
            return self.get('Omega_g') + self.get('Omega_ur') + self._like(self.get('Omega_pncdm_tot'))
        if name == 'N_ncdm':
            return len(params['m_ncdm'])
        if name == 'm_ncdm_tot':
            return sum(params['m_ncdm']) if params['m_ncdm'] else 0.
        if name == 'T_ncdm':
            return np.array(params['T_ncdm_over_cmb']) * params['T_cmb'] if not _is_array(params['T_cmb']) else [t * params['T_cmb'] for t in params['T_ncdm_over_cmb']]
        if name in ('Omega_ncdm', 'Omega_pncdm'):   # today's density / 3 x pressure of every species over rho_crit (cosmology.py:371-376)
            rck = bgmod_constants()[2] / (1e10 * 1.98847 * 1e30) * (1e6 * 3.085677581491367e16)**3
            fac = 1. if name == 'Omega_ncdm' else 3.
            if not params['m_ncdm']:      # no species: nothing to bring to the host (a device-to-host copy of h would stall the queue)
                return np.array([])
            T_cmb, h = np.asarray(_host(params['T_cmb']), dtype='f8'), np.asarray(_host(params['h']), dtype='f8')
            vals = [fac * _ncdm_momenta_z0(T_cmb * t, m, 'rho' if name == 'Omega_ncdm' else 'p') / h**2 / rck for t, m in zip(params['T_ncdm_over_cmb'], params['m_ncdm'])]
            return np.array(vals) if all(np.ndim(v) == 0 for v in vals) else vals
        if name in ('Omega_ncdm_tot', 'Omega_pncdm_tot'):
            vals = self.get(name[:-4])
            return sum(vals) if len(vals) else 0.
        if name == 'Omega_m':
            return params['Omega_b'] + params['Omega_cdm'] + self._like(self.get('Omega_ncdm_tot')) - self._like(self.get('Omega_pncdm_tot'))
        if name == 'Omega_de':
            return 1. - (params['Omega_cdm'] + params['Omega_b'] + self.get('Omega_g') + self.get('Omega_ur') + self._like(self.get('Omega_ncdm_tot'))
                         + params['Omega_k'])
        if name == 'Omega_Lambda':
            return 0. if self._has_fld else self.get('Omega_de')
        if name == 'Omega_fld':
            return self.get('Omega_de') if self._has_fld else 0.
        if name == 'K':
            return - 100.**2 / (c / 1e3)**2 * params['Omega_k']
        if name == 'theta_cosmomc':   # sound horizon over angular distance at last scattering, CosmoMC approximation (cosmology.py:202-228, 404-408)
            ba = self.get_background()
            omega_b, omega_m = self.get('omega_b'), self.get('omega_m')
            zstar = 1048 * (1 + 0.00124 * omega_b**(-0.738)) * (1 + (0.0783 * omega_b**(-0.238) / (1 + 39.5 * omega_b**0.763))
                                                               * omega_m**(0.560 / (1 + 21.1 * omega_b**1.81)))
            rs = ba._eval_per_cosmology('rs_cosmomc', zstar)
            if bool((rs != rs).any()):
                raise CosmologyComputationError('precision not achieved in the sound horizon integral')
            return rs * _host(ba.h) / ba._eval_per_cosmology('comoving_transverse_distance', zstar)
        if name == 'theta_MC_100':
            return self.get('theta_cosmomc') * 100.
        if name == 'N_eff':   # cosmology.py:402-403
            return sum(t**4 * (4. / 11.)**(-4. / 3.) for t in params['T_ncdm_over_cmb']) + params['N_ur']
        return _missing

    def _like(self, v):
        """Host value -> same kind as the (possibly torch) parameters it is combined with."""
        if np.ndim(v) and any(dv.is_torch(x) for x in self._params.values()):
            ref = next(x for x in self._params.values() if dv.is_torch(x))
            return dv.upload(np.asarray(v, dtype='f8'), ref.device, cache=False)
        return v

    @property
    def _has_fld(self):
        p = self._params
        if any(_is_array(p[n]) for n in ('w0_fld', 'wa_fld', 'cs2_fld')):
            return True
        return (p['w0_fld'] != -1) or (p['wa_fld'] != 0) or (p['cs2_fld'] != 1.)

    @property
    def batch_size(self):
        """Number of cosmologies when parameters are arrays, else None."""
        if '_batch_size' not in self.__dict__:      # parameters do not change after construction
            self.__dict__['_batch_size'] = self._find_batch_size()
        return self.__dict__['_batch_size']

    def _find_batch_size(self):
        for name, v in self._params.items():
            if name in _default_calculation_parameters:   # z_pk, modes ...: arrays that are not batches
                continue
            for x in (v if name in ('m_ncdm', 'T_ncdm_over_cmb') else [v]):
                if _is_array(x):
                    return int(np.size(x)) if not dv.is_torch(x) else int(x.numel())
        return None

    def bg_params(self):
        """The background parameter block of the kernels; with massive neutrinos also their density / pressure tables, under 'ncdm'."""
        bg = {name: self._params[name] for name in _lib.BG_PARAMS}
        ncdm = self.ncdm_tables()
        if ncdm is not None:
            bg['ncdm'] = ncdm
        return bg

    def ncdm_tables(self):
        """:class:`cosmoprimo_amd.background.NcdmTables` of the cosmology (the splines of density and pressure of every massive species the
        reference's background builds, cosmology.py:1961-1998), None without species; built on the device once and shared by the sections and
        engines of one cosmology."""
        params = self._params
        if not params.get('m_ncdm'):
            return None
        memo = self.__dict__.setdefault('_derived_memo', {})
        if '_ncdm_tables' not in memo:
            device = getattr(self, 'device', None)
            if device is None:
                device = dv.resolve_device(getattr(self, '_device', None), *[v for v in params.values() if not isinstance(v, (list, tuple))])
            memo['_ncdm_tables'] = bgmod.NcdmTables(params['m_ncdm'], params['T_ncdm_over_cmb'], h=params['h'], T_cmb=params['T_cmb'],
                                                    ncosmo=self.batch_size or 1, device=device)
        return memo['_ncdm_tables']


def bgmod_constants():
    c, sb = 299792458.0, 5.6703744191844314e-08   # scipy.constants.c, Stefan_Boltzmann
    mpc = 1e6 * 3.085677581491367e16
    rck = 3.0 * (100. * 1e3 / mpc)**2 / (8 * np.pi * 6.6743e-11)   # constants.rho_crit_over_kgph_per_mph3
    return c, sb, rck


# the reference's list (cosmology.py:13); no engine of this package has the Perturbations / Harmonic sections (Boltzmann codes only): their getters
# exist and raise what the reference's raise for an engine without the section
_Sections = ['Background', 'Thermodynamics', 'Primordial', 'Perturbations', 'Transfer', 'Harmonic', 'Fourier']


class RegisteredEngine(type):

    """Metaclass registering :class:`BaseEngine`-derived classes by their ``name`` (reference cosmology.py:459-468)."""
    _registry = {}

    def __new__(meta, name, bases, class_dict):
        cls = super().__new__(meta, name, bases, class_dict)
        meta._registry[cls.name] = cls
        return cls


class BaseEngine(BaseCosmoParams, metaclass=RegisteredEngine):

    """Base engine for cosmological calculation (reference cosmology.py:471-571)."""
    name = 'base'

    def __init__(self, cosmo, device=None, **extra_params):
        self._params = dict(cosmo._params)
        # derived parameters depend on the parameters only: the engines of one cosmology (and the cosmology) keep them in one place -- a second
        # engine (the no-wiggle template of a filter, say) then derives nothing again (each derivation is a few tiny device kernels for a batch)
        self._derived_memo = cosmo.__dict__.setdefault('_derived_memo', {})
        self._extra_params = extra_params
        self._rsigma8 = None
        self.device = dv.resolve_device(device if device is not None else getattr(cosmo, '_device', None), *self._params.values())
        self._Sections = {}
        module = sys.modules[self.__class__.__module__]
        for name in _Sections:   # sections are module-level classes of the engine's module, discovered by name (reference :497-502)
            Section = getattr(module, name, None)
            if Section is not None:
                self._Sections[name.lower()] = Section
        self._sections = {}

    def _get_A_s_fid(self):
        """First guess for A_s given sigma8 (reference cosmology.py:505-510)."""
        if 'A_s' in self._params:
            return self._params['A_s']
        return 2.43e-9 * (self['sigma8'] / 0.87659)**2

    def _rescale_sigma8(self):
        """Rescale perturbative quantities to match input sigma8 (reference eisenstein_hu.py:94-103)."""
        if getattr(self, '_rsigma8', None) is not None:
            return self._rsigma8
        self._rsigma8 = 1.
        if 'sigma8' in self._params and self._normalise_batch_on_device():
            return self._rsigma8
        if 'sigma8' in self._params:
            fo = self.get_fourier()
            sigma8 = self['sigma8']
            s8m = fo._sigma8_m_device()
            self._rsigma8 = dv.to_device(sigma8, self.device) / s8m
            if self.batch_size is None:
                self._rsigma8 = float(self._rsigma8)
            self._sections = {name: section for name, section in self._sections.items() if name in ('background', 'thermodynamics')}   # untouched by the rescaling
        return self._rsigma8

    def _normalise_batch_on_device(self):
        """The sigma8 normalisation of a batch of cosmologies of an analytic engine as one kernel (``cp_sigma8_normalise``): the factors, the
        normalised amplitudes and the normalised spectra on the 1024 wavenumbers every sigma integral and every filter asks for next, left where
        :meth:`pk_params` and the Fourier section look for them.  False (nothing done) for anything else."""
        transfer = getattr(self, '_transfer', None)
        if self.batch_size is None or transfer not in _lib.ENGINES or getattr(self.device, 'type', None) != 'cuda':
            return False
        from .interpolator import sigma8_normalise
        res = sigma8_normalise(transfer, self.bg_params(), self.pk_params(rsigma8=1.), self['sigma8'], self.device)
        if res is None:
            return False
        rsigma8, amplitude, spectra, k = res
        self._rsigma8 = rsigma8
        self.__dict__['_A_s_normalised'] = (rsigma8, amplitude)
        self.__dict__['_pk0_normalised'] = ((k.shape, k.tobytes()), rsigma8, spectra)
        return True


def _make_section_getter(section):

    def getter(self):
        name = section.lower()
        if name not in self._sections:
            self._sections[name] = self._Sections[name](self)    # KeyError for a section the engine's module does not define, as the reference
        return self._sections[name]

    getter.__name__ = 'get_{}'.format(section.lower())
    getter.__doc__ = """Return :class:`{}` calculations (reference cosmology.py:557-571).""".format(section)
    return getter


for section in _Sections:
    setattr(BaseEngine, 'get_{}'.format(section.lower()), _make_section_getter(section))


def get_engine(engine):
    """Return the engine class for a name / class (reference cosmology.py:574-633)."""
    if isinstance(engine, str):
        engine = engine.lower()
        if engine not in RegisteredEngine._registry:
            raise CosmologyError('Unknown engine {}; available on the MI355X path: {}'.format(engine, sorted(n for n in RegisteredEngine._registry if n != 'base')))
        return RegisteredEngine._registry[engine]
    return engine


class Cosmology(BaseCosmoParams):

    """Cosmology, defined as a set of parameters (and possibly a current engine attached to it) (reference cosmology.py:724-1477)."""
    _default_cosmological_parameters = _default_cosmological_parameters
    _default_calculation_parameters = _default_calculation_parameters
    _conflict_parameters_no_alias = _conflict_parameters_no_alias
    _alias_parameters = _alias_parameters
    _conflict_parameters = _conflict_parameters

    def __init__(self, engine=None, extra_params=None, device=None, **params):
        check_params(params, conflicts=_conflict_parameters)
        self._derived = {}
        self._engine = None
        self._device = device
        self._input_params = merge_params(self.get_default_params(include_conflicts=False), params, conflicts=_conflict_parameters)
        self._params = _compile_params(self._input_params)
        if engine is not None:
            self.set_engine(engine, **(extra_params or {}))

    @property
    def engine(self):
        return self._engine

    def set_engine(self, engine, set_engine=True, **extra_params):
        """Set engine for cosmological calculation (reference cosmology.py:636-668, 1219-1235)."""
        if isinstance(engine, BaseEngine):
            new = engine
        else:
            new = get_engine(engine)(self, **extra_params)
        if set_engine:
            self._engine = new
        return new

    def clone(self, base='input', engine=None, extra_params=None, **params):
        r"""
        Copy with updated engine and parameters (reference cosmology.py:1237-1290).  ``base='input'``: update the input parameters (with input
        :math:`h, \omega_b, \omega_{cdm}`, a new ``h`` keeps the physical densities); ``base='internal'`` (or None): update the compiled
        :math:`h, \Omega_b, \Omega_{cdm}` basis (a new ``h`` keeps the density parameters).
        """
        check_params(params, conflicts=_conflict_parameters)
        if base == 'input':
            base_params = dict(self._input_params)
        elif base in ('internal', None):
            base_params = dict(self._params)
        else:
            raise CosmologyInputError('Unknown parameter base {}'.format(base))
        if engine is None and self._engine is not None:
            engine = self._engine.__class__
        if extra_params is None:   # the current engine's, if the engine class is unchanged
            same = engine is not None and self._engine is not None and get_engine(engine).name == self._engine.name
            extra_params = dict(self._engine._extra_params) if same else {}
        return self.__class__(engine=engine, extra_params=extra_params, device=self._device, **merge_params(base_params, params, conflicts=_conflict_parameters))

    def solve(self, param, func, target=0., limits=None, init=None, xtol=1e-6, maxiter=25):
        """
        Cosmology with ``func(cosmo) == target``, varying input parameter ``param`` (one scalar cosmology; reference cosmology.py:1292-1376).

        func : callable on a :class:`Cosmology`, or a parameter name: 'theta_MC_100' as in the reference (any name ``cosmo[name]`` knows works).
        limits : bracket for ``param``; else the bracket is searched from ``init`` = x0 or (x0, dx) (default: the current value; for 'h' / 'H0'
        matched to 'theta_MC_100' the fitting formula of class_public's shooting gives the start).  xtol : absolute tolerance on ``param``.
        """
        from scipy import optimize
        if func is None:
            raise CosmologyInputError('Provide func')
        if self.batch_size is not None:
            raise NotImplementedError('solve() varies one parameter of one cosmology; call it per cosmology of a batch')
        name = func if isinstance(func, str) else None
        if name is not None:
            def func(cosmo):
                return cosmo[name]

        def f(value):
            try:
                return float(np.asarray(_host(func(self.clone(base='input', **{param: value}))))) - target
            except CosmologyError:
                raise ValueError('cosmology could not be computed for {} = {}'.format(param, value))

        def fail(exc, limits):
            values = []
            for x in limits:
                try:
                    values.append(f(x) + target)
                except ValueError:
                    values.append(np.nan)
            raise CosmologyInputError('Could not find proper {} value in the interval that matches target = {:.4f} with [f({:.3f}), f({:.3f})] = [{:.4f}, {:.4f}]'
                                      .format(param, target, *limits, *values)) from exc

        if limits is None:
            scale = {'h': 1., 'H0': 100.}.get(param, None)
            if init is None and name == 'theta_MC_100' and scale is not None:
                init = (scale * (3.54 * target**2 - 5.455 * target + 2.548), scale * 0.02)
            if init is None:
                init = self[param]
            if np.ndim(init) == 0:
                if scale is None:
                    raise ValueError('provide either init tuple (x0, dx) = (initial value, typical variation), or parameter limits')
                init = (init, 0.1 * scale)
            x0, dx = float(init[0]), abs(float(init[1]))
            # widen [x0 - w, x0 + w] until f changes sign; a side where the cosmology cannot be computed stops growing
            lo = hi = x0
            flo = fhi = f(x0)
            found = flo == 0.
            for it in range(maxiter):
                if found:
                    break
                for side in (-1, 1):
                    x = (lo if side < 0 else hi) + side * dx
                    try:
                        fx = f(x)
                    except ValueError:
                        continue
                    if side < 0:
                        if fx * flo <= 0.:
                            lo, hi, found = x, lo, True
                            break
                        lo, flo = x, fx
                    else:
                        if fx * fhi <= 0.:
                            lo, hi, found = hi, x, True
                            break
                        hi, fhi = x, fx
                dx *= 1.5
            if not found:
                fail(ValueError('no sign change'), (lo, hi))
            limits = (lo, hi)
        limits = tuple(float(x) for x in limits)
        try:
            value = optimize.brentq(f, *limits, xtol=xtol, rtol=4 * np.finfo(float).eps, maxiter=max(4 * maxiter, 100))
        except (ValueError, RuntimeError) as exc:
            fail(exc, limits)
        return self.clone(base='input', **{param: value})

    def __getstate__(self):
        """State dictionary: compiled, input and derived parameters, engine name and extra parameters (reference cosmology.py:1390-1397)."""
        state = {'engine': None}
        for name in ['params', 'input_params', 'derived']:
            state[name] = getattr(self, '_{}'.format(name))
        if getattr(self, '_engine', None) is not None:
            state['engine'] = {'name': self._engine.name, 'extra_params': self._engine._extra_params}
        return state

    def __setstate__(self, state):
        for name in ['params', 'input_params', 'derived']:
            setattr(self, '_{}'.format(name), state.get(name, {}))
        self._device, self._engine = None, None
        if state.get('engine', None) is not None:
            self.set_engine(state['engine']['name'], **state['engine']['extra_params'])

    @classmethod
    def from_state(cls, state):
        new = cls.__new__(cls)
        new.__setstate__(state)
        return new

    @classmethod
    def read(cls, filename):
        """Read from disk: '.json' or numpy '.npy' (reference cosmology.py:1406-1416)."""
        import json
        filename = str(filename)
        if filename.endswith('.json'):
            with open(filename, 'r') as file:
                state = _from_json(json.load(file))
        else:
            state = np.load(filename, allow_pickle=True)[()]
        return cls.from_state(state)

    def write(self, filename):
        """Write to disk: '.json' or numpy '.npy' (reference cosmology.py:1425-1434); batches held on the device are written as arrays."""
        import json
        import os
        filename = str(filename)
        dirname = os.path.dirname(filename)
        if dirname:
            os.makedirs(dirname, exist_ok=True)
        state = _to_json(self.__getstate__())
        if filename.endswith('.json'):
            with open(filename, 'w') as file:
                json.dump(state, file)
        else:
            np.save(filename, _from_json(state), allow_pickle=True)

    @classmethod
    def load(cls, filename):
        import warnings
        warnings.warn('load() is deprecated, use read() instead.', DeprecationWarning, stacklevel=2)
        return cls.read(filename)

    def save(self, filename):
        import warnings
        warnings.warn('save() is deprecated, use write() instead.', DeprecationWarning, stacklevel=2)
        return self.write(filename)

    @_class_or_instancemethod
    def get_default_params(self=None, of=None, include_conflicts=True):
        """Class or instance method: with an engine set, its defaults are added (reference cosmology.py:823-847)."""
        toret = BaseCosmoParams.get_default_params(of=of, include_conflicts=include_conflicts)
        engine = getattr(self, '_engine', None) if isinstance(self, Cosmology) else None
        if engine is not None:
            toret.update(engine.get_default_params(of=of, include_conflicts=include_conflicts))
        return toret

    @_class_or_instancemethod
    def get_default_parameters(self=None):
        """Deprecated name of :meth:`get_default_params` (reference cosmology.py:848-852)."""
        import warnings
        warnings.warn('get_default_parameters is deprecated, use get_default_params')
        return Cosmology.get_default_params() if self is None or not isinstance(self, Cosmology) else self.get_default_params()

    def get_params(self, of='base'):
        toret = super().get_params(of=of)
        if self._engine is not None:
            toret.update(self._engine.get_params(of=of))
        return toret

    def __dir__(self):
        """Members plus those found in exactly one section of the engine (reference cosmology.py:1442-1457)."""
        toret = list(super().__dir__())
        if self._engine is None:
            return toret
        for Section in self._engine._Sections.values():
            for item in _section_dir(Section):
                if item in toret:
                    toret.remove(item)
                else:
                    toret.append(item)
        return toret

    def __getattr__(self, name):
        """``cosmo.get_<section>()``, and the attributes of the engine's sections when one section only has them:
        ``cosmo.comoving_radial_distance`` is ``cosmo.get_background().comoving_radial_distance`` (reference cosmology.py:1459-1473)."""
        if name.startswith('__') or name in ('_engine', '_params', '_input_params', '_derived', '_device'):
            raise AttributeError(name)
        if self._engine is None:
            raise AttributeError('Attribute {} not found; try setting an engine ("set_engine")?'.format(name))
        Sections = self._engine._Sections
        owners = [section_name for section_name, Section in Sections.items() if name in _section_dir(Section)]
        if len(owners) == 1:
            return getattr(getattr(self._engine, 'get_{}'.format(owners[0]))(), name)
        raise AttributeError("Attribute {} not found in any of {} engine's products (rejecting duplicates)".format(name, self._engine.__class__.__name__))

    def __eq__(self, other):
        return type(other) == type(self) and _deepeq(other._params, self._params) and other._engine == self._engine

    __hash__ = object.__hash__


def _section_dir(Section):
    """Public names a section class offers (per-instance quantities are properties of the class: ``utils.addproperty``)."""
    return [item for item in dir(Section) if not item.startswith('_')]


def _get_cosmology_engine(cosmology, engine=None, set_engine=True, **extra_params):
    """The engine of ``cosmology``: its current one (``engine`` None), or a new one, attached if ``set_engine`` (reference cosmology.py:636-668)."""
    if engine is None:
        if cosmology._engine is None:
            raise CosmologyInputError('Please provide an engine')
        return cosmology._engine
    return cosmology.set_engine(engine, set_engine=set_engine, **extra_params)


def _get_section(cosmology, section, engine=None, set_engine=True, **extra_params):
    return getattr(_get_cosmology_engine(cosmology, engine=engine, set_engine=set_engine, **extra_params), 'get_' + section)()


def _make_section_getter(section):

    def getter(cosmology, engine=None, set_engine=True, **extra_params):
        return _get_section(cosmology, section.lower(), engine=engine, set_engine=set_engine, **extra_params)

    getter.__name__ = section
    getter.__doc__ = 'Return :class:`{}` calculations of ``cosmology`` for ``engine`` (reference cosmology.py:671-705).'.format(section)
    return getter


for section in _Sections:    # Background(cosmology, engine=None, set_engine=True, **extra_params), ..., Harmonic(...)
    globals()[section] = _make_section_getter(section)


def _make_section_getter(section):

    def getter(self, engine=None, set_engine=True, **extra_params):
        engine = _get_cosmology_engine(self, engine=engine, set_engine=set_engine, **extra_params)
        toret = getattr(engine, 'get_{}'.format(section), None)
        if toret is None:
            raise CosmologyInputError('Engine {} does not provide {}'.format(engine.__class__.__name__, section))
        return toret()

    getter.__name__ = 'get_{}'.format(section)
    getter.__doc__ = 'Get {} (reference cosmology.py:1509-1540).'.format(section)
    return getter


for section in _Sections:    # Cosmology.get_background(engine=None, set_engine=True, **extra_params), ...
    setattr(Cosmology, 'get_{}'.format(section.lower()), _make_section_getter(section.lower()))


class MetaSection(type(object)):

    """Metaclass of the sections (reference cosmology.py:1480-1487: there it registers them as pytree nodes; nothing to register here)."""

    def __new__(meta, name, bases, class_dict):
        return super().__new__(meta, name, bases, class_dict)


class BaseSection(object, metaclass=MetaSection):

    """Base section (reference cosmology.py:1480-1540)."""

    def __init__(self, engine):
        self._engine = engine    # an engine, or (DefaultBackground only, as in the reference's tests) a Cosmology without one
        self.device = engine.device if hasattr(engine, 'device') else dv.resolve_device(getattr(engine, '_device', None), *engine._params.values())
        self._h = engine['h']

    @property
    def engine(self):
        return self._engine


def _out(t, like, dtype=None):
    """Device tensor -> numpy (default) or torch if ``like`` is a torch tensor."""
    dtype = dtype if dtype is not None else dv.float_dtype(like)
    return _finish(t, dtype, dv.is_torch(like))


# today's quantities, stored per instance as '_name' (the reference's utils.addproperty list, cosmology.py:1627-1630)
@utils.addproperty('H0', 'h', 'N_ur', 'N_ncdm', 'm_ncdm', 'm_ncdm_tot', 'N_eff', 'T0_cmb', 'T0_ncdm', 'w0_fld', 'wa_fld', 'cs2_fld',
                   'Omega0_cdm', 'Omega0_b', 'Omega0_k', 'K', 'Omega0_g', 'Omega0_ur', 'Omega0_r',
                   'Omega0_pncdm', 'Omega0_pncdm_tot', 'Omega0_ncdm', 'Omega0_ncdm_tot',
                   'Omega0_m', 'Omega0_Lambda', 'Omega0_fld', 'Omega0_de')
class BaseBackground(BaseSection):

    """Background densities, E(z) and distances (reference BaseBackground, cosmology.py:1627-1933), evaluated by ``cp_background_distance``."""

    def __init__(self, engine):
        super().__init__(engine)
        for name in ['H0', 'h', 'N_ur', 'N_ncdm', 'm_ncdm', 'm_ncdm_tot', 'N_eff', 'w0_fld', 'wa_fld', 'cs2_fld', 'K']:
            setattr(self, '_{}'.format(name), engine[name])
        self._T0_cmb = engine['T_cmb']
        for name in ['cdm', 'b', 'k', 'g', 'ur', 'r', 'ncdm', 'pncdm', 'ncdm_tot', 'pncdm_tot', 'm', 'Lambda', 'fld', 'de']:
            setattr(self, '_Omega0_{}'.format(name), engine['Omega_{}'.format(name)])
        self._bg = {name: engine._params[name] for name in _lib.BG_PARAMS}
        self._T0_ncdm = engine['T_ncdm']
        self._ncdm = engine.ncdm_tables()   # massive neutrinos: spline tables of density and pressure, built on the device once (cosmology.py:1961-1998)

    def _eval(self, kind, z, species=None):
        return bgmod.distance(kind, z, self._bg, device=self.device, ncdm=self._ncdm, species=species)

    def efunc(self, z):
        r"""E(z) = H(z) / H0, unitless (cosmology.py:1751-1754)."""
        return self._eval('efunc', z)

    def hubble_function(self, z):
        """Hubble function, in km/s/Mpc (cosmology.py:1756-1759)."""
        return self._eval('hubble_function', z)

    def rho_crit(self, z):
        """Comoving critical density excluding curvature, in 1e10 Msun/h / (Mpc/h)^3 (cosmology.py:1738-1749)."""
        return self._eval('rho_crit', z)

    def Omega_m(self, z):
        """Density parameter of matter at z (cosmology.py:1796)."""
        return self._eval('Omega_m', z)

    def Omega_de(self, z):
        """Density parameter of dark energy at z (cosmology.py:1850)."""
        return self._eval('Omega_de', z)

    def rs(self, z):
        """Comoving sound horizon at z, in Mpc/h (cosmology.py:1914-1933): the reference's fixed-depth Romberg rule (15 refinements) of
        c_s dtau/da from a = 1e-8, one wave per sample.  Like the reference it fails (CosmologyComputationError) where that rule misses
        its 1e-7 tolerance, e.g. at z = 0."""
        out = self._eval('rs', z)
        if bool((out != out).any()):
            raise CosmologyComputationError('precision not achieved in the sound horizon integral')
        return out

    def _eval_per_cosmology(self, kind, z):
        """``kind`` at one redshift per cosmology (z: float, or (B,) for a batch)."""
        if self._engine.batch_size is None:
            return self._eval(kind, z)
        zt = dv.to_device(z, self.device).reshape(-1, 1)
        return bgmod.distance(kind, zt, self._bg, device=self.device, ncdm=self._ncdm, per_cosmology_z=True)[:, 0].cpu().numpy()

    def rho_tot(self, z):
        """Comoving total density (matter + radiation + dark energy), in 1e10 Msun/h / (Mpc/h)^3 (cosmology.py:1731-1736)."""
        return self._eval('rho_tot', z)

    def T_cmb(self, z):
        """CMB temperature at z, in K (cosmology.py:1762-1764)."""
        return self._eval('T_cmb', z)

    def _zeros_like(self, z):
        return self._eval('rho_g', z) * 0.

    def _de_split(self, which, form, z):
        """rho_Lambda / rho_fld (cosmology.py:1714-1722): dark energy is either a cosmological constant or a fluid (Cosmology._has_fld)."""
        has_fld = self._engine._has_fld
        if np.ndim(has_fld):
            raise NotImplementedError('{}(z) for a batch mixing cosmological-constant and fluid dark energy'.format(which))
        if bool(has_fld) == (form == 'fld'):
            return self._eval(which, z)
        return self._zeros_like(z)

    def rho_Lambda(self, z):
        """Comoving density of the cosmological constant (cosmology.py:1714-1717)."""
        return self._de_split('rho_Lambda', 'Lambda', z)

    def rho_fld(self, z):
        """Comoving density of the dark energy fluid (cosmology.py:1719-1722)."""
        return self._de_split('rho_fld', 'fld', z)

    def Omega_Lambda(self, z):
        """Density parameter of the cosmological constant at z (cosmology.py:1841-1844)."""
        return self._de_split('Omega_Lambda', 'Lambda', z)

    def Omega_fld(self, z):
        """Density parameter of the dark energy fluid at z (cosmology.py:1846-1849)."""
        return self._de_split('Omega_fld', 'fld', z)

    # massive neutrinos: interpolated tables as DefaultBackground (cosmology.py:1961-1998); identically zero without massive species
    def _per_species(self, kind, z, species, scale=1.):
        """One species (int): z.shape; several (None = all, or a sequence of indices): (nspecies,) + z.shape."""
        if species is not None and np.ndim(species) == 0:
            return self._eval(kind, z, species=int(species)) * scale if self._N_ncdm else self._zeros_like(z)
        species = list(range(self._N_ncdm)) if species is None else [int(s) for s in species]
        if not species:
            first = self._zeros_like(z)
            return np.zeros((0,) + tuple(first.shape), dtype=first.dtype) if not dv.is_torch(first) else first.new_zeros((0,) + tuple(first.shape))
        vals = [self._eval(kind, z, species=s) * scale for s in species]
        return dv.torch().stack(vals) if dv.is_torch(vals[0]) else np.stack(vals)

    def _ncdm_tot(self, kind, z, scale=1.):
        return self._eval(kind, z) * scale if self._N_ncdm else self._zeros_like(z)

    def rho_ncdm(self, z, species=None):
        """Comoving density of massive neutrinos, every species (N_ncdm,) + z.shape or one (cosmology.py:1961-1978)."""
        return self._per_species('rho_ncdm', z, species)

    def p_ncdm(self, z, species=None):
        """Pressure of massive neutrinos (cosmology.py:1980-1998)."""
        return self._per_species('p_ncdm', z, species)

    def rho_ncdm_tot(self, z):
        """Total comoving density of massive neutrinos (cosmology.py:1661-1663)."""
        return self._ncdm_tot('rho_ncdm', z)

    def p_ncdm_tot(self, z):
        """Total pressure of massive neutrinos (cosmology.py:1675-1677)."""
        return self._ncdm_tot('p_ncdm', z)

    def Omega_ncdm(self, z, species=None):
        """Density parameter of massive neutrinos at z (cosmology.py:1812-1819)."""
        return self._per_species('Omega_ncdm', z, species)

    def Omega_ncdm_tot(self, z):
        """Total density parameter of massive neutrinos at z (cosmology.py:1821-1824)."""
        return self._ncdm_tot('Omega_ncdm', z)

    def Omega_pncdm(self, z, species=None):
        """Density parameter of the pressure of massive neutrinos, 3 p / rho_crit (cosmology.py:1826-1833)."""
        return self._per_species('pfrac_ncdm', z, species, scale=3)

    def Omega_pncdm_tot(self, z):
        """Total density parameter of the pressure of massive neutrinos (cosmology.py:1835-1838)."""
        return self._ncdm_tot('pfrac_ncdm', z, scale=3)

    def T_ncdm(self, z, species=None):
        """Temperature of the massive neutrinos, K: (N_ncdm,) + z.shape, or one species (cosmology.py:1766-1772)."""
        zp1 = self._eval('T_cmb', z) / self._T0_cmb
        T0 = np.asarray(self._T0_ncdm, dtype='f8')
        if species is not None and np.ndim(species) == 0:
            return zp1 * float(T0[species])
        T0 = T0 if species is None else T0[[int(s) for s in species]]
        if dv.is_torch(zp1):
            return dv.torch().as_tensor(T0, device=zp1.device, dtype=zp1.dtype).reshape((-1,) + (1,) * zp1.ndim) * zp1
        return T0.astype(zp1.dtype).reshape((-1,) + (1,) * np.ndim(zp1)) * zp1

    _use_table_spline = True      # (tests switch it off to compare with the per-sample kernel)

    def _table_spline(self, kind):
        """The reference's own representation of D_C(z) / time(z) (cosmology.py:2011, 2036-2042): the natural cubic spline through the values at
        the interpolation knots.  The background kernel evaluates that spline per (cosmology, z) sample from a fresh quadrature -- the right
        shape for batches of cosmologies; for ONE cosmology (a few redshifts or a catalogue) the table is computed once (one launch on the knots)
        and the spline evaluated per point (``cp_spline_points``): same numbers to rounding, 237 E(z) evaluations per redshift less."""
        cache = self.__dict__.setdefault('_table_splines', {})
        if kind not in cache:
            from .interpolator import Interpolator1D
            knots = np.empty(400 if kind == 'time' else 119)
            _lib.check(_lib.load().cp_background_knots(_lib.as_double_p(knots), knots.size))
            cache[kind] = Interpolator1D(knots, self._eval(kind, knots), k=3, extrap=False, assume_sorted=True, device=self.device)
            cache[kind]._npoints_operator = 0      # always point by point: no operator per set of redshifts
        return cache[kind]

    def comoving_radial_distance(self, z):
        """Comoving radial distance, in Mpc/h (cosmology.py:2027-2042)."""
        if self._use_table_spline and self._engine.batch_size is None:     # one cosmology: its table once, then a cubic per redshift (a few or a catalogue of them)
            return self._table_spline('comoving_radial_distance')(z)
        return self._eval('comoving_radial_distance', z)

    def _curved_distance(self, kind, z):
        """Angular / transverse / luminosity distance of ONE cosmology from its radial-distance table (cosmology.py:1855-1912): the curvature
        map and the powers of 1 + z applied on the device in float64, in the kernel's operation order; dtype and container of ``z`` kept."""
        zt = dv.to_device(z, self.device).contiguous()
        chi = self._table_spline('comoving_radial_distance')(zt).contiguous()       # float64 device tensor
        if chi.numel():      # one pass, in place (cp_distance_from_radial: the last lines of the background kernel)
            _lib.check(_lib.load().cp_distance_from_radial(chi.data_ptr(), zt.data_ptr(), chi.numel(), float(self._K), _lib.BG_KINDS[kind], chi.data_ptr(),
                                                           self.device.index, dv.stream_of(self.device)))
        return _out(chi, z)

    def _distance(self, kind, z):
        if self._use_table_spline and self._engine.batch_size is None:
            return self._curved_distance(kind, z)
        return self._eval(kind, z)

    def angular_diameter_distance(self, z):
        """Proper angular diameter distance, in Mpc/h (cosmology.py:1855-1868)."""
        return self._distance('angular_diameter_distance', z)

    def comoving_transverse_distance(self, z):
        """Comoving transverse distance, in Mpc/h (cosmology.py:1893-1900)."""
        return self._distance('comoving_transverse_distance', z)

    comoving_angular_distance = comoving_transverse_distance

    def luminosity_distance(self, z):
        """Luminosity distance, in Mpc/h (cosmology.py:1904-1912)."""
        return self._distance('luminosity_distance', z)

    def angular_diameter_distance_2(self, z1, z2):
        """Angular diameter distance of an object at z2 seen from z1 (cosmology.py:1870-1890); scalar cosmologies."""
        if np.any(_host(z2) < _host(z1)):
            import warnings
            warnings.warn('Second redshift(s) z2 ({}) is less than first redshift(s) z1 ({}).'.format(z2, z1))
        torch = dv.torch()
        like, dtype = (z1 if dv.is_torch(z1) else z2), dv.float_dtype(z1, z2)
        chi = dv.to_device(self.comoving_radial_distance(dv.to_device(z2, self.device)), self.device) - \
            dv.to_device(self.comoving_radial_distance(dv.to_device(z1, self.device)), self.device)
        K = dv.to_device(self._K, self.device)
        K = K.reshape(K.shape + (1,) * (chi.ndim - K.ndim))
        sq = torch.sqrt(torch.abs(K))
        safe = torch.where(sq > 0, sq, torch.ones_like(sq))
        sk = torch.where(K > 0, torch.sin(safe * chi) / safe, torch.where(K < 0, torch.sinh(safe * chi) / safe, chi))
        return _out(sk / (1 + dv.to_device(z2, self.device)), like, dtype)


def _add_density_methods():
    docs = {'g': 'photons', 'b': 'baryons', 'ur': 'massless neutrinos', 'cdm': 'cold dark matter', 'k': 'curvature',
            'r': 'radiation (photons + massless neutrinos)', 'm': 'matter (cdm + baryons)', 'de': 'dark energy (fluid + cosmological constant)'}

    def make(kind, doc):
        def method(self, z):
            return self._eval(kind, z)
        method.__name__ = kind
        method.__doc__ = doc
        return method

    for name, what in docs.items():
        setattr(BaseBackground, 'rho_' + name, make('rho_' + name, 'Comoving density of {}, in 1e10 Msun/h / (Mpc/h)^3 (cosmology.py:1680-1729).'.format(what)))
        if name not in ('m', 'de'):    # Omega_m(z), Omega_de(z) are defined above
            setattr(BaseBackground, 'Omega_' + name, make('Omega_' + name, 'Density parameter of {} at z, unitless (cosmology.py:1774-1853).'.format(what)))


_add_density_methods()


class DefaultBackground(BaseBackground):

    """Background with the quantities that need a quadrature or an ODE (reference DefaultBackground, cosmology.py:1954-2093): interpolated
    massive neutrinos, time / age, comoving radial distance (in :class:`BaseBackground` here: same kernels) and the linear growth from
    its ODE.  The analytic engines override the growth with closed forms (``eisenstein_hu.Background``), as in the reference."""

    def time(self, z):
        """Proper time (age of the universe at z), in Gyr (reference DefaultBackground.time, cosmology.py:2000-2012): the RK4 scan on the
        400-knot grid and its natural spline, evaluated by the same device kernel as the distances; NaN outside [0, 1e8 - 1]."""
        return self._eval('time', z)

    @property
    def age(self):
        """Current age of the universe, in Gyr (cosmology.py:2014-2025); one value per cosmology."""
        return self._eval('age', np.zeros(()))

    def _growth_tables(self, mass):
        if mass not in ('m', 'cb'):
            raise ValueError("mass must be one of ['m', 'cb']")
        cache = self.__dict__.setdefault('_growth_cache', {})
        if mass not in cache:   # 200 RK4 steps per cosmology on the device (cp_growth_ode_tables), once per section
            cache[mass] = bgmod.growth_ode_tables(self._bg, mass=mass, ncdm=self._ncdm, ncosmo=self._engine.batch_size or 1, device=self.device)
        return cache[mass]

    def _growth_eval(self, which, z, mass):
        """Natural spline (as Interpolator1D) through row ``which`` (0: D, 1: D'/D) of the ODE tables at z and at 0: device (ncosmo, nq + 1)."""
        from .spline import LinearOperator
        knots, tab = self._growth_tables(mass)
        zq = np.concatenate([_host(z).ravel(), [0.]])
        return LinearOperator.spline(knots, zq, bc='natural', device=self.device)(tab[:, which].contiguous())

    def growth_factor(self, z, mass='m', znorm=None):
        """Linear growth factor from its ODE in ln a (reference cosmology.py:2044-2087): normalised to 1 at z = 0, or to (1 + znorm) / (1 + z)
        deep in matter domination if ``znorm`` is given; ``mass``: 'm' or 'cb'.  NaN outside the solved range z <= e^6 - 1."""
        out = self._growth_eval(0, z, mass)
        out = (1. + znorm) * out[:, :-1] if znorm is not None else out[:, :-1] / out[:, -1:]
        zshape = tuple(np.shape(_host(z)))
        shape = ((out.shape[0],) if self._engine.batch_size is not None else ()) + zshape
        return _out(out.reshape(shape), z)

    def growth_rate(self, z, mass='m'):
        """Linear growth rate d ln D / d ln a from the same ODE (reference cosmology.py:2089-2094)."""
        out = self._growth_eval(1, z, mass)[:, :-1]
        zshape = tuple(np.shape(_host(z)))
        shape = ((out.shape[0],) if self._engine.batch_size is not None else ()) + zshape
        return _out(out.reshape(shape), z)
