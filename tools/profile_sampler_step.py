"""cProfile of one sampler step (fresh Cosmology -> sigma8_z, distances, rs_drag, wallish2018 pknow): where the host time goes.  python tools/profile_sampler_step.py"""
import cProfile
import os
import pstats
import sys
import warnings

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main(top=40):
    import torch
    import cosmoprimo_amd as cp
    from cosmoprimo_amd.bao_filter import PowerSpectrumBAOFilter
    warnings.simplefilter('ignore')
    zq = np.array([0., 0.5, 1.])
    state = {'i': 0}

    def step():
        state['i'] += 1
        c = cp.Cosmology(engine='eisenstein_hu', Omega_m=0.3 + 1e-4 * (state['i'] % 50), sigma8=0.8)
        p = c.get_fourier().pk_interpolator()
        out = [np.asarray(p.sigma8_z(0.)), np.asarray(c.get_background().comoving_radial_distance(zq)), np.asarray(c.get_thermodynamics().rs_drag)]
        out.append(np.asarray(PowerSpectrumBAOFilter(p.to_1d(z=0.), engine='wallish2018').pknow))
        torch.cuda.synchronize()
        return out

    for _ in range(5):
        step()
    prof = cProfile.Profile()
    prof.enable()
    for _ in range(20):
        step()
    prof.disable()
    pstats.Stats(prof).sort_stats('cumulative').print_stats(top)


if __name__ == '__main__':
    main()
