"""Constants under the reference's names (cosmoprimo/constants.py): scipy.constants plus the cosmological conversion factors the host code
and the kernels (csrc/cp_cosmo_common.h) use."""
from scipy.constants import *   # noqa: F401,F403
from scipy import constants as _sc

electronvolt_over_joule = 1.602176634e-19
megaparsec_over_m = 1e6 * _sc.parsec
msun_over_kg = 1.98847e30
rho_crit_over_kgph_per_mph3 = 3.0 * (100. * 1e3 / megaparsec_over_m)**2 / (8 * _sc.pi * _sc.gravitational_constant)                # h^2 kg / m^3
rho_crit_over_Msunph_per_Mpcph3 = rho_crit_over_kgph_per_mph3 / (1e10 * msun_over_kg) * megaparsec_over_m**3   # 1e10 Msun/h / (Mpc/h)^3
TNCDM_OVER_CMB = 0.71611     # temperature of massive neutrinos over the CMB's, as CLASS
NEFF = 3.044
TCMB = 2.7255
gigayear_over_megaparsec = 3.06601394e2
