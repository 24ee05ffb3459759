"""Config 4 (BAO filters on batches of EH P(k) vectors) alone, for rocprofv3 --kernel-trace --stats:  python tools/profile_config4.py [wallish2018|brieden2022] [nvectors]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    import cosmoprimo_amd as cp
    engine = sys.argv[1] if len(sys.argv) > 1 else 'wallish2018'
    nb = int(sys.argv[2]) if len(sys.argv) > 2 else 32768
    dev = torch.device('cuda', 0)
    rng = np.random.default_rng(2)
    par = dict(Omega_m=rng.uniform(.25, .40, nb), Omega_b=rng.uniform(.04, .06, nb), h=rng.uniform(.6, .8, nb), n_s=rng.uniform(.92, 1., nb))
    fid = cp.Cosmology(engine='eisenstein_hu')
    for rep in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        stamps = []
        for start in range(0, nb, 16384):
            sl = slice(start, min(nb, start + 16384))
            cosmo = cp.Cosmology(engine='eisenstein_hu', sigma8=0.8, **{k: torch.as_tensor(v[sl], device=dev) for k, v in par.items()})
            interp = cosmo.get_fourier().pk_interpolator(z=np.array([0.]))
            torch.cuda.synchronize(); t1 = time.perf_counter()
            kw = dict(cosmo=cosmo, cosmo_fid=fid) if engine != 'wallish2018' else {}
            f = cp.PowerSpectrumBAOFilter(interp, engine=engine, **kw)
            torch.cuda.synchronize(); t2 = time.perf_counter()
            stamps.append((t1, t2))
        dt = time.perf_counter() - t0
        filt = sum(b - a for a, b in stamps)
        print('%s: %d vectors in %.1f ms (%.3g vectors/s); filter part %.1f ms' % (engine, nb, dt * 1e3, nb / dt, filt * 1e3))


if __name__ == '__main__':
    main()
