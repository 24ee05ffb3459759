"""Config 3 (sigma_rz 256 r x 64 z for 10 000 EH cosmologies) alone, for rocprofv3 --kernel-trace --stats:  python tools/profile_config3.py [ncosmo]"""
import os
import sys
import time
import warnings

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    import cosmoprimo_amd as cp
    warnings.simplefilter('ignore')
    nb = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
    dev = torch.device('cuda', 0)
    rng = np.random.default_rng(1)
    par = dict(Omega_m=rng.uniform(.25, .40, nb), Omega_b=rng.uniform(.04, .06, nb), h=rng.uniform(.6, .8, nb), n_s=rng.uniform(.92, 1., nb), sigma8=0.8)
    r, z = np.geomspace(1, 100, 256), np.linspace(0, 3, 64)
    cosmo = cp.Cosmology(engine='eisenstein_hu', **{k: (torch.as_tensor(v, device=dev) if np.ndim(v) else v) for k, v in par.items()})
    interp = cosmo.get_fourier().pk_interpolator()
    rt, zt = torch.as_tensor(r, device=dev), torch.as_tensor(z, device=dev)
    interp.sigma_rz(rt, zt)
    torch.cuda.synchronize()
    for rep in range(3):
        t0 = time.perf_counter()
        out = interp.sigma_rz(rt, zt)
        torch.cuda.synchronize()
        print('sigma_rz: %d cosmologies in %.2f ms' % (nb, (time.perf_counter() - t0) * 1e3), tuple(out.shape))


if __name__ == '__main__':
    main()
