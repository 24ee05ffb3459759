"""MI355X-native implementation of cosmoprimo's FFTLog / P(k) hot path (same API names as cosmoprimo)."""
from .fftlog import (FFTlog, HankelTransform, PowerToCorrelation, CorrelationToPower, TophatVariance, GaussianVariance, pad,
                     BesselJKernel, SphericalBesselJKernel, TophatKernel, TophatSqKernel, GaussianKernel, GaussianSqKernel)

__version__ = '0.1.0'
