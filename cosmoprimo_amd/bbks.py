"""BBKS engine on MI355X (reference cosmoprimo/bbks.py; the polynomial is reproduced as coded there)."""
import warnings

from .eisenstein_hu import EisensteinHuEngine, Background, Primordial, Fourier  # noqa: F401
from .eisenstein_hu import Transfer as _Transfer


class BBKSEngine(EisensteinHuEngine):
    """BBKS no-wiggle analytic formulae (reference bbks.py:10-38)."""
    name = 'bbks'
    _transfer = 'bbks'

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        if self['N_ncdm']:      # bbks.py:24-31: these warnings are live in the reference
            warnings.warn('{} cannot cope with massive neutrinos'.format(self.__class__.__name__))
        if self.batch_size is None:
            if self['Omega_k'] != 0.:
                warnings.warn('{} cannot cope with non-zero curvature'.format(self.__class__.__name__))
            if self._has_fld:
                warnings.warn('{} cannot cope with non-constant dark energy'.format(self.__class__.__name__))


class Transfer(_Transfer):
    """BBKS matter transfer function, the polynomial as the reference codes it (bbks.py:41-64: ``3.89 q (16.2 q)^2``): the engine's
    ``_transfer = 'bbks'`` selects ``CP_ENGINE_BBKS`` of ``cp_power_eval``."""
