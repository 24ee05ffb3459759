"""The sigma8 normalisation of a batch of analytic cosmologies (one radius, one redshift; eisenstein_hu.py:94-103): the fused kernel (P(k) -> FFTLog ->
spline in LDS) against the dot product of the spectrum with the functional of the same pipeline (cp_sigma_rz_functional), alternately in one
process.   python tools/bench_sigma8.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch      # noqa: E402
from cosmoprimo_amd import interpolator as itp      # noqa: E402

dev = torch.device('cuda', 0)
rng = np.random.default_rng(2)
n = 16384
Om, Ob, h, ns = rng.uniform(.25, .40, n), rng.uniform(.04, .06, n), rng.uniform(.6, .8, n), rng.uniform(.92, 1., n)
bg = dict(h=torch.as_tensor(h, device=dev), Omega_cdm=torch.as_tensor(Om - Ob, device=dev), Omega_b=torch.as_tensor(Ob, device=dev))
pk = dict(n_s=torch.as_tensor(ns, device=dev))
g2 = torch.ones((n, 1), dtype=torch.float64, device=dev)
r = np.array([8.])


def run(radii, engine):
    itp._FUNCTIONAL_RADII = radii
    for _ in range(5):
        itp.sigma_rz_analytic(engine, bg, pk, r, g2, dev, keep_spectra=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        itp.sigma_rz_analytic(engine, bg, pk, r, g2, dev, keep_spectra=True)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / 20 * 1e3


for engine in ('eisenstein_hu', 'eisenstein_hu_nowiggle', 'bbks'):
    for rep in range(2):
        print('%-24s fused kernel %.3f ms   functional %.3f ms' % (engine, run(0, engine), run(4, engine)))
