"""cProfile of one single-cosmology filter construction + pknow (host-side cost of the filters): python tools/profile_filter_latency.py peakaverage [n]"""
import cProfile
import os
import pstats
import sys
import warnings

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main(engine, top=35):
    import torch
    import cosmoprimo_amd as cp
    warnings.simplefilter('ignore')
    cosmo = cp.Cosmology(engine='eisenstein_hu')
    pk1d = cosmo.get_fourier().pk_interpolator().to_1d(z=0.)
    kwargs = dict(cosmo=cosmo, cosmo_fid=cosmo) if engine in ('peakaverage', 'brieden2022', 'ehpoly', 'hinton2017') else {}

    def run():
        out = np.asarray(cp.PowerSpectrumBAOFilter(pk1d, engine=engine, **kwargs).pknow)
        torch.cuda.synchronize()
        return out

    for _ in range(3):
        run()
    prof = cProfile.Profile()
    prof.enable()
    for _ in range(5):
        run()
    prof.disable()
    pstats.Stats(prof).sort_stats('cumulative').print_stats(top)


if __name__ == '__main__':
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 35)
