"""MI355X-native implementation of cosmoprimo's FFTLog / P(k) hot path (same API names as cosmoprimo)."""
from .fftlog import (FFTlog, HankelTransform, PowerToCorrelation, CorrelationToPower, TophatVariance, GaussianVariance, pad,
                     BesselJKernel, SphericalBesselJKernel, TophatKernel, TophatSqKernel, GaussianKernel, GaussianSqKernel)
from .interpolator import (PowerSpectrumInterpolator1D, PowerSpectrumInterpolator2D, CorrelationFunctionInterpolator1D,
                           CorrelationFunctionInterpolator2D)
from .cosmology import (Cosmology, Background, Thermodynamics, Primordial, Transfer, Fourier, CosmologyError, CosmologyInputError,
                        CosmologyComputationError)
from . import eisenstein_hu, eisenstein_hu_nowiggle, eisenstein_hu_nowiggle_variants, bbks, tabulated  # noqa: F401  (registers the engines)
from .bao_filter import PowerSpectrumBAOFilter, CorrelationFunctionBAOFilter
from . import fiducial, constants  # noqa: F401

__version__ = '0.1.0'
