"""Host-side profile (cProfile) of config 4: where the Python time of one 16384-cosmology chunk goes.  python tools/hostprofile_config4.py [wallish2018|brieden2022]"""
import cProfile
import os
import pstats
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    import cosmoprimo_amd as cp
    engine = sys.argv[1] if len(sys.argv) > 1 else 'wallish2018'
    nb = 16384
    dev = torch.device('cuda', 0)
    rng = np.random.default_rng(2)
    par = dict(Omega_m=rng.uniform(.25, .40, nb), Omega_b=rng.uniform(.04, .06, nb), h=rng.uniform(.6, .8, nb), n_s=rng.uniform(.92, 1., nb))
    fid = cp.Cosmology(engine='eisenstein_hu')

    filters = {}

    def chunk():
        cosmo = cp.Cosmology(engine='eisenstein_hu', sigma8=0.8, **{k: torch.as_tensor(v, device=dev) for k, v in par.items()})
        interp = cosmo.get_fourier().pk_interpolator(z=np.array([0.]))
        kw = dict(cosmo=cosmo, cosmo_fid=fid) if engine != 'wallish2018' else {}
        if engine not in filters:
            filters[engine] = cp.PowerSpectrumBAOFilter(interp, engine=engine, **kw)
        else:
            filters[engine](interp, cosmo=cosmo if kw else None)
        out = filters[engine].pknow
        torch.cuda.synchronize()
        return out

    chunk()
    chunk()
    prof = cProfile.Profile()
    prof.enable()
    for _ in range(4):
        chunk()
    prof.disable()
    st = pstats.Stats(prof)
    st.sort_stats('cumulative').print_stats(45)
    st.sort_stats('tottime').print_stats(25)
    st.print_callers('to_host')
    st.print_callers("'cpu' of")


if __name__ == '__main__':
    main()
