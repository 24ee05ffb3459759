"""Eisenstein & Hu no-wiggle engine on MI355X (reference cosmoprimo/eisenstein_hu_nowiggle.py)."""
from .eisenstein_hu import EisensteinHuEngine, Background, Thermodynamics, Primordial, Fourier  # noqa: F401
from .eisenstein_hu import Transfer as _Transfer


class EisensteinHuNoWiggleEngine(EisensteinHuEngine):
    """Eisenstein & Hu no-wiggle analytic formulae (reference eisenstein_hu_nowiggle.py:7-21)."""
    name = 'eisenstein_hu_nowiggle'
    _transfer = 'eisenstein_hu_nowiggle'


class Transfer(_Transfer):
    """No-wiggle matter transfer function (reference eisenstein_hu_nowiggle.py:24-51): the engine's ``_transfer = 'eisenstein_hu_nowiggle'``
    selects ``CP_ENGINE_EH_NOWIGGLE`` of ``cp_power_eval``."""
