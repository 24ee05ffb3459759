"""
Fiducial cosmologies under the reference's names (cosmoprimo/fiducial.py): parameter sets of published analyses.  The values are data
(Planck 2018 papers, BOSS DR12, the AbacusSummit table, the Uchuu simulations); engines are this package's ('eisenstein_hu', ..., default
'eisenstein_hu_nowiggle_variants', the analytic engine that handles the massive species these cosmologies have) instead of the reference's
Boltzmann codes.
"""
import os

from .cosmology import Cosmology, TNCDM_OVER_CMB, NEFF

_DEFAULT_ENGINE = 'eisenstein_hu_nowiggle_variants'


def Planck2018FullFlatLCDM(engine=_DEFAULT_ENGINE, extra_params=None, **params):
    """Planck 2018 TT, TE, EE, lowE, lensing and BAO, flat LCDM (reference fiducial.py:50-71)."""
    default_params = dict(h=0.6766, omega_cdm=0.11933, omega_b=0.02242, Omega_k=0., sigma8=0.8102, k_pivot=0.05, n_s=0.9665, m_ncdm=[0.06],
                          neutrino_hierarchy=None, T_ncdm_over_cmb=TNCDM_OVER_CMB, N_eff=NEFF, tau_reio=0.0561, A_L=1.0, w0_fld=-1., wa_fld=0.)
    return Cosmology(engine=engine, extra_params=extra_params, **default_params).clone(**params)


def BOSS(engine=_DEFAULT_ENGINE, extra_params=None, **params):
    """BOSS DR12 fiducial cosmology, arXiv:1607.03155 (reference fiducial.py:74-98)."""
    default_params = dict(h=0.676, Omega_m=0.31, omega_b=0.022, Omega_k=0., sigma8=0.8, k_pivot=0.05, n_s=0.97, m_ncdm=[0.06],
                          neutrino_hierarchy=None, T_ncdm_over_cmb=TNCDM_OVER_CMB, N_eff=NEFF, A_L=1.0, w0_fld=-1., wa_fld=0.)
    return Cosmology(engine=engine, extra_params=extra_params, **default_params).clone(**params)


_AbacusSummit_params_filename = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'data', 'abacus_cosmologies.json')


def AbacusSummit_params(name=None, filename=_AbacusSummit_params_filename, params=None):
    """
    AbacusSummit cosmological parameters (https://github.com/abacusorg/AbacusSummit/tree/master/Cosmologies; reference fiducial.py:108-155):
    the dictionary of cosmology ``name`` (e.g. ``0`` or ``'000'``), or the list for every cosmology; ``params``: the names to return, default
    ['omega_b', 'omega_cdm', 'h', 'A_s', 'n_s', 'alpha_s', 'N_ur', 'omega_ncdm', 'omega_k', 'tau_reio', 'w0_fld', 'wa_fld'] ('root' is accepted).
    The table is kept as JSON (``filename``), one entry per cosmology.
    """
    import json
    import re
    if name is not None and not isinstance(name, str):
        name = '{:03d}'.format(name)
    if params is None:
        params = ['omega_b', 'omega_cdm', 'h', 'A_s', 'n_s', 'alpha_s', 'N_ur', 'omega_ncdm', 'omega_k', 'tau_reio', 'w0_fld', 'wa_fld']
    default = {'tau_reio': 0.0544, 'omega_k': 0.}     # not in the table
    with open(filename, 'r') as file:
        table = json.load(file)
    toret = []
    for root, row in table.items():
        tmp = {name_: value for name_, value in default.items() if name_ in params}      # these two come first, as in the reference
        for param in params:
            if param in default:
                continue
            value = root if param == 'root' else row[param]
            tmp[param] = tuple(value) if isinstance(value, list) else value
        if name is not None:
            if re.match('[^0-9]*{}$'.format(name), root):
                return tmp
        else:
            toret.append(tmp)
    if name is not None:
        raise ValueError('AbacusSummit cosmology {} not found'.format(name))
    return toret


def AbacusSummit(name=0, engine=_DEFAULT_ENGINE, precision=None, extra_params=None, **params):
    """AbacusSummit cosmology ``name`` (reference fiducial.py:158-228).  As in the reference, ``N_ur`` is recast into ``N_eff`` so that later
    changes of the neutrino masses are continuous.  ``precision`` (settings of the Boltzmann code there) is accepted and unused."""
    default_params = dict(k_pivot=0.05, neutrino_hierarchy=None, T_ncdm_over_cmb=TNCDM_OVER_CMB, A_L=1.0)
    default_params.update(AbacusSummit_params(name=name))
    cosmo = Cosmology(engine=engine, extra_params=extra_params, **default_params)
    cosmo = cosmo.clone(base='input', N_eff=cosmo['N_eff'])
    return cosmo.clone(**params)


def AbacusSummitBase(engine=_DEFAULT_ENGINE, precision=None, extra_params=None, **params):
    """Base AbacusSummit cosmology (reference fiducial.py:231-254)."""
    return AbacusSummit(name='000', engine=engine, precision=precision, extra_params=extra_params, **params)


DESI = AbacusSummitBase


def DESIDR2Flatw0waCDM(engine=_DEFAULT_ENGINE, precision=None, extra_params=None, **params):
    """Best fit of flat w0waCDM to CMB + DESI DR2 BAO + DES-Y5 supernovae, arXiv:2503.14738 (reference fiducial.py:295-327): these values on
    top of the AbacusSummit base cosmology."""
    bestfit_params = {'Omega_m': 0.3191980194, 'omega_b': 0.02221485621, 'H0': 66.73428704, 'logA': 3.038847745, 'n_s': 0.9644215278,
                      'tau_reio': 0.05271118001, 'w0_fld': -0.7536302620, 'wa_fld': -0.8574714585}
    return AbacusSummit(engine=engine, precision=precision, extra_params=extra_params, **bestfit_params).clone(**params)


def Uchuu(name='Planck2015', engine=_DEFAULT_ENGINE, extra_params=None, **params):
    """Cosmologies of the Uchuu simulations (reference fiducial.py:11-47)."""
    common = dict(Omega_k=0., m_ncdm=[0.06], neutrino_hierarchy=None, T_ncdm_over_cmb=TNCDM_OVER_CMB, N_eff=NEFF, A_L=1.0, k_pivot=0.05)
    table = {'Planck2015': dict(h=0.6774, Omega_m=0.3089, Omega_b=0.0486, sigma8=0.8159, n_s=0.9667, tau_reio=0.063),
             'Planck2018': dict(h=0.6766, Omega_m=0.3111, Omega_b=0.048975, sigma8=0.8102, n_s=0.9665, tau_reio=0.063),
             'Planck2018DDE': dict(h=0.6766, Omega_m=0.3111, Omega_b=0.048975, sigma8=0.8102, n_s=0.9665, tau_reio=0.063, w0_fld=-0.45, wa_fld=-1.79),
             'DESIY1DDE': dict(h=0.6470, Omega_m=0.3440, Omega_b=0.048975, sigma8=0.8102, n_s=0.9665, tau_reio=0.063, w0_fld=-0.45, wa_fld=-1.79)}
    if name not in table:
        raise NotImplementedError('Uchuu cosmology {} not implemented; available cosmologies are {}'.format(name, list(table)))
    return Cosmology(engine=engine, extra_params=extra_params, **table[name], **common).clone(**params)


_desi_table = {}


def _tabulate_DESI():
    """(z, E(z), D_C(z) [Mpc/h]) of the DESI fiducial cosmology on the reference's grid, z = [0] + logspace(-8, 2, 40001) (fiducial.py:285-291).
    E(z) comes from this package's background kernels (massive neutrino included); D_C(z) is the cumulated 8-point Gauss-Legendre integral of
    c / (100 E) over every table interval -- not the 119-knot spline of ``comoving_radial_distance``, whose end condition costs 1e-3 below z = 0.1."""
    import numpy as np
    if not _desi_table:
        ba = DESI().get_background()
        z = np.concatenate([[0.], np.logspace(-8, 2, 40001)])
        x, w = np.polynomial.legendre.leggauss(8)
        lo, hi = z[:-1], z[1:]
        nodes = 0.5 * (hi - lo)[:, None] * (x[None, :] + 1.) + lo[:, None]
        integrand = 299792.458 / (100. * ba.efunc(nodes.ravel()).reshape(nodes.shape))
        steps = 0.5 * (hi - lo) * (integrand * w[None, :]).sum(axis=1)
        _desi_table.update(z=z, efunc=ba.efunc(z), comoving_radial_distance=np.concatenate([[0.], np.cumsum(steps)]))
    return _desi_table


def TabulatedDESI():
    """
    Tabulated DESI cosmology (reference fiducial.py:272-282): E(z) and D_C(z) interpolated linearly in a 40 002-row table over 0 <= z <= 100,
    for whole catalogues of redshifts at once.  The reference ships the table as a data file computed with a Boltzmann code; here it is computed
    on first use from :func:`DESI` (same parameters; E(z) within 2e-6 and D_C within 3e-6 of the reference's file, tests/test_fiducial_gpu.py).
    """
    return DESI(engine='tabulated', extra_params={'table': _tabulate_DESI()})


def save_TabulatedDESI(filename):
    """Write the table of :func:`TabulatedDESI` in the reference's file format (fiducial.py:285-291): readable by its 'tabulated' engine."""
    import numpy as np
    table = _tabulate_DESI()
    header = 'z = [0] + np.logspace(-8, 2, 40001)\nz efunc(z) comoving_radial_distance(z) [Mpc/h]'
    np.savetxt(filename, np.array([table['z'], table['efunc'], table['comoving_radial_distance']]).T, fmt='%.18e', header=header, comments='# ')
