"""Eisenstein & Hu no-wiggle engine on MI355X (reference cosmoprimo/eisenstein_hu_nowiggle.py)."""
from .eisenstein_hu import EisensteinHuEngine, Background, Thermodynamics, Primordial, Transfer, Fourier  # noqa: F401


class EisensteinHuNoWiggleEngine(EisensteinHuEngine):
    """Eisenstein & Hu no-wiggle analytic formulae (reference eisenstein_hu_nowiggle.py:7-21)."""
    name = 'eisenstein_hu_nowiggle'
    _transfer = 'eisenstein_hu_nowiggle'
