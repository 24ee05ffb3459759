"""
CosmoSIS module: ``setup`` / ``execute`` / ``cleanup`` over :class:`cosmoprimo_amd.Cosmology` (reference
bindings/cosmosis/cosmoprimo_interface.py:12-222).  Same option names, same DataBlock sections, entries and units as the reference's module,
so a pipeline's ini file switches by pointing ``file =`` at this one; the engines are this package's (analytic P(k) on the GPU), hence no CMB
spectra: ``harmonic = T`` raises.

The module is written as three tables -- how DataBlock entries become ``Cosmology`` parameters, which distance-like arrays go to the
``distances`` section, which growth arrays go to ``growth_parameters`` -- walked by ``execute``.
"""
import sys
import traceback
import warnings

import numpy as np

try:        # section names come from the framework when it is there; they are plain strings
    from cosmosis.datablock import names as _names, option_section
    COSMO, DISTANCES, GROWTH = _names.cosmological_parameters, _names.distances, _names.growth_parameters
except ImportError:
    option_section, COSMO, DISTANCES, GROWTH = 'module_options', 'cosmological_parameters', 'distances', 'growth_parameters'

from .. import constants
from ..cosmology import Cosmology, CosmologyError, get_engine

# option -> (getter of the options object, default)
OPTIONS = {'zmin': ('get_double', 0.0), 'zmax': ('get_double', 3.01), 'nz': ('get_int', 150), 'lmax': ('get_int', 2000), 'kmax': ('get_double', 50.0),
           'debug': ('get_bool', False), 'harmonic': ('get_bool', False), 'lensing': ('get_bool', True), 'fourier': ('get_bool', False),
           'nonlinear': ('get_string', ''), 'engine': ('get_string', 'eisenstein_hu')}

# Cosmology parameter -> (section, entry, transform); entries that must be there
REQUIRED = {'A_s': (COSMO, 'A_s', None), 'n_s': (COSMO, 'n_s', None), 'H0': (COSMO, 'h0', lambda h: 100. * h), 'omega_b': (COSMO, 'ombh2', None),
            'omega_cdm': (COSMO, 'omch2', None), 'Omega_k': (COSMO, 'omega_k', None), 'tau_reio': (COSMO, 'tau', None)}
# ... with a default
DEFAULTED = {'T_cmb': (COSMO, 'TCMB', 2.726), 'N_eff': (COSMO, 'nnu', 3.046)}
# ... passed on only when the pipeline set them
OPTIONAL = {'alpha_s': (COSMO, 'nrun'), 'w0_fld': (COSMO, 'w'), 'wa_fld': (COSMO, 'wa'), 'cs2_fld': (COSMO, 'cs2_de'), 'A_L': (COSMO, 'A_lens'),
            'reionization_width': ('reionization', 'delta_redshift'), 'YHe': (COSMO, 'YHe')}
DISTANCE_STEP = 0.01


def setup(options):
    """Options of the ini file, fixed along the chain; ``cosmoprimo_<name>`` options are forwarded to :class:`Cosmology` as ``<name>``."""
    config = {name: getattr(options, getter)(option_section, name, default=default) for name, (getter, default) in OPTIONS.items()}
    for _, key in options.keys(option_section):
        if key.startswith('cosmoprimo_'):
            config[key] = options[option_section, key]
    return config


def cosmology_parameters(block, config):
    """The ``Cosmology`` arguments for the current point of the chain."""
    engine = get_engine(config.get('engine', 'eisenstein_hu'))
    params = {'engine': engine, 'lensing': bool(config['harmonic'] and config['lensing']), 'non_linear': config['nonlinear'],
              'use_ppf': config.get('use_ppf', True)}
    for name, (section, entry, transform) in REQUIRED.items():
        value = block[section, entry]
        params[name] = value if transform is None else transform(value)
    for name, (section, entry, default) in DEFAULTED.items():
        params[name] = block.get_double(section, entry, default=default)
    # neutrinos: one mass `mnu` per massive species, or the total mass shared out by the hierarchy
    nmassive = block.get_int(COSMO, 'num_massive_neutrinos', default=None)
    mnu = block.get_double(COSMO, 'mnu', default=0.06)
    if nmassive is None or nmassive == 3:
        params['m_ncdm'], params['neutrino_hierarchy'] = mnu, block.get_string(COSMO, 'neutrino_hierarchy', default=None)
    else:
        params['m_ncdm'], params['neutrino_hierarchy'] = [mnu] * nmassive, None
    for name, where in OPTIONAL.items():
        if block.has_value(*where):
            params[name] = block[where]
    translated = {entry for table in (REQUIRED, DEFAULTED, OPTIONAL) for section, entry, *_ in table.values() if section == COSMO}
    translated |= {'mnu', 'num_massive_neutrinos', 'neutrino_hierarchy', 'massless_nu', 'omega_nu', 'omnuh2'}      # read above / ignored below
    for name in engine.get_default_params(include_conflicts=True):      # any other parameter of the engine, given under its own name
        if name not in translated and block.has_value(COSMO, name):
            params[name] = block[COSMO, name]
    if config['harmonic']:
        params['ellmax_cl'] = config['lmax']
    if config['fourier']:
        params['z_pk'] = np.linspace(config['zmin'], config['zmax'], config['nz'])
    if block.has_value(COSMO, 'massless_nu'):
        warnings.warn('massless_nu is ignored: set nnu, the effective number of relativistic species in the early Universe')
    if (block.has_value(COSMO, 'omega_nu') or block.has_value(COSMO, 'omnuh2')) and not block.has_value(COSMO, 'mnu'):
        warnings.warn('omega_nu and omnuh2 are ignored: set mnu and num_massive_neutrinos instead')
    params.update({key[len('cosmoprimo_'):]: value for key, value in config.items() if key.startswith('cosmoprimo_')})
    return params


def _fourier_outputs(block, cosmo, ba):
    fo = cosmo.get_fourier()
    z = np.asarray(cosmo['z_pk'], dtype='f8')
    for section, of in (('matter_power_lin', 'delta_m'), ('cdm_baryon_power_lin', 'delta_cb')):
        interp = fo.pk_interpolator(of=of)
        block.put_grid(section, 'k_h', interp.k, 'z', interp.z, 'p_k', interp.pk)
    if cosmo['non_linear']:
        interp = fo.pk_interpolator(of='delta_m', non_linear=True)
        block.put_grid('matter_power_nl', 'k_h', interp.k, 'z', interp.z, 'p_k', interp.pk)
    sigma8 = {of: np.asarray(fo.sigma8_z(z, of=of)) for of in ('delta_m', 'delta_cb', 'theta_cb')}
    sigma8_today = float(fo.sigma8_z(0., of='delta_m'))
    growth = {'z': z, 'a': 1. / (1. + z), 'sigma_8': sigma8['delta_m'], 'fsigma_8': sigma8['theta_cb'], 'd_z': sigma8['delta_m'] / sigma8_today,
              'f_z': sigma8['theta_cb'] / sigma8['delta_cb']}
    for name, value in growth.items():
        block[GROWTH, name] = value
    block[COSMO, 'sigma_8'] = sigma8_today
    block[COSMO, 'sigma_12'] = float(fo.sigma_rz(12. / ba.h, 0., of='delta_m'))      # 12 Mpc, not Mpc/h
    block[COSMO, 'S_8'] = sigma8_today * np.sqrt(ba.Omega0_m / 0.3)


def _distance_outputs(block, cosmo, ba, config):
    z = np.arange(config['zmin'], config['zmax'] + DISTANCE_STEP, DISTANCE_STEP)
    h = ba.h
    lum, ang = np.asarray(ba.luminosity_distance(z)), np.asarray(ba.angular_diameter_distance(z))      # Mpc/h
    transverse = ang * (1. + z)
    hubble = 100. * np.asarray(ba.efunc(z)) / (constants.c / 1e3)                                        # h/Mpc
    volume = (z * transverse**2 / hubble)**(1. / 3.)
    rs_drag = cosmo.get_thermodynamics().rs_drag
    modulus = np.full_like(lum, -np.inf)
    positive = lum > 0
    modulus[positive] = 5. * np.log10(lum[positive]) + 25.      # as the reference: of the distance in Mpc/h
    out = {'z': z, 'nz': len(z), 'D_L': lum / h, 'D_A': ang / h, 'D_M': transverse / h, 'D_V': volume / h, 'H': hubble * h, 'MU': modulus,
           'age': ba.age, 'rs_zdrag': rs_drag / h, 'rs_DV': rs_drag * volume, 'F_AP': transverse * hubble}
    for name, value in out.items():
        block[DISTANCES, name] = value


def execute(block, config):
    """One point of the chain: DataBlock -> Cosmology -> DataBlock.  Returns 0, or 1 when the cosmology cannot be computed."""
    try:
        if config['harmonic']:
            raise CosmologyError('harmonic = T asks for CMB spectra, which need a Boltzmann code: the engines of cosmoprimo_amd are the analytic ones')
        cosmo = Cosmology(**cosmology_parameters(block, config))
        ba = cosmo.get_background()
        if config['fourier']:
            _fourier_outputs(block, cosmo, ba)
        _distance_outputs(block, cosmo, ba, config)
    except CosmologyError as error:
        if config['debug']:
            sys.stderr.write('Error in cosmoprimo_amd. You set debug=T so here is more debug info:\n')
            traceback.print_exc(file=sys.stderr)
        else:
            sys.stderr.write('Error in cosmoprimo_amd. Set debug=T for info: {}\n'.format(error))
        return 1
    return 0


def cleanup(config):
    return 0
