"""MI355X-native implementation of cosmoprimo's FFTLog / P(k) hot path (same API names as cosmoprimo)."""
from .fftlog import (FFTlog, HankelTransform, PowerToCorrelation, CorrelationToPower, TophatVariance, GaussianVariance, pad,
                     BesselJKernel, SphericalBesselJKernel, TophatKernel, TophatSqKernel, GaussianKernel, GaussianSqKernel)
from .interpolator import (PowerSpectrumInterpolator1D, PowerSpectrumInterpolator2D, CorrelationFunctionInterpolator1D,
                           CorrelationFunctionInterpolator2D)
from .cosmology import (Cosmology, Background, Thermodynamics, Primordial, Perturbations, Transfer, Harmonic, Fourier, CosmologyError,
                        CosmologyInputError, CosmologyComputationError)
from . import eisenstein_hu, eisenstein_hu_nowiggle, eisenstein_hu_nowiggle_variants, bbks, tabulated  # noqa: F401  (registers the engines)
from .bao_filter import PowerSpectrumBAOFilter, CorrelationFunctionBAOFilter
from . import fiducial, constants  # noqa: F401

# the reference's list (cosmoprimo/__init__.py:7-11): what ``from cosmoprimo import *`` gives
__all__ = ['Cosmology', 'Background', 'Thermodynamics', 'Primordial', 'Transfer', 'Harmonic', 'Fourier', 'CosmologyError']
__all__ += ['PowerSpectrumInterpolator1D', 'PowerSpectrumInterpolator2D', 'CorrelationFunctionInterpolator1D', 'CorrelationFunctionInterpolator2D']
__all__ += ['FFTlog', 'PowerToCorrelation', 'CorrelationToPower', 'TophatVariance']
__all__ += ['PowerSpectrumBAOFilter', 'CorrelationFunctionBAOFilter']
__all__ += ['fiducial']

__version__ = '0.1.0'
