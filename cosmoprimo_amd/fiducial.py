"""
Fiducial cosmologies under the reference's names (cosmoprimo/fiducial.py): parameter sets of published analyses.  The values are data
(Planck 2018 papers, BOSS DR12, the AbacusSummit table, the Uchuu simulations); engines are this package's ('eisenstein_hu', ..., default
'eisenstein_hu_nowiggle_variants', the analytic engine that handles the massive species these cosmologies have) instead of the reference's
Boltzmann codes.
"""
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


# AbacusSummit c000 (https://github.com/abacusorg/AbacusSummit/tree/master/Cosmologies): Planck 2018 base_plikHM_TTTEEE_lowl_lowE_lensing mean
_ABACUS_000 = dict(omega_b=0.02237, omega_cdm=0.1200, h=0.6736, A_s=2.0830e-9, n_s=0.9649, alpha_s=0.0, N_ur=2.0328, omega_ncdm=(0.00064420,),
                   omega_k=0., tau_reio=0.0544, w0_fld=-1.0, wa_fld=0.0)


def AbacusSummit(name=0, engine=_DEFAULT_ENGINE, precision=None, extra_params=None, **params):
    """AbacusSummit cosmology ``name`` (reference fiducial.py:158-228); only the base cosmology c000 is tabulated here.  As in the
    reference, ``N_ur`` is recast into ``N_eff`` so that later changes of the neutrino masses are continuous."""
    if not isinstance(name, str):
        name = '{:03d}'.format(name)
    if name != '000':
        raise NotImplementedError('AbacusSummit cosmology {} is not tabulated here (only the base cosmology 000)'.format(name))
    cosmo = Cosmology(engine=engine, extra_params=extra_params, **_ABACUS_000)
    cosmo = cosmo.clone(base='input', N_eff=cosmo['N_eff'])
    return cosmo.clone(**params)


def AbacusSummitBase(engine=_DEFAULT_ENGINE, precision=None, extra_params=None, **params):
    """Base AbacusSummit cosmology (reference fiducial.py:231-254)."""
    return AbacusSummit(name='000', engine=engine, precision=precision, extra_params=extra_params, **params)


DESI = AbacusSummitBase


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
