"""power_kernel timing against the number of wavenumbers per cosmology (is the per-cosmology or the per-k part the cost?)  python tools/time_power_kernel.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    from cosmoprimo_amd import power
    dev = torch.device('cuda', 0)
    rng = np.random.default_rng(1)
    for engine in ['eisenstein_hu', 'eisenstein_hu_nowiggle', 'bbks']:
        for nb in [10000, 100]:
            bg = dict(Omega_cdm=torch.as_tensor(rng.uniform(.2, .35, nb), device=dev), Omega_b=torch.as_tensor(rng.uniform(.04, .06, nb), device=dev),
                      h=torch.as_tensor(rng.uniform(.6, .8, nb), device=dev))
            pk = dict(A_s=2e-9, n_s=torch.as_tensor(rng.uniform(.92, 1., nb), device=dev))
            for nk in [256, 1024, 4096, 16384]:
                k = torch.as_tensor(np.geomspace(1e-5, 1e2, nk), device=dev)
                for what, z in [('matter', None), ('matter', np.linspace(0., 3., 30)), ('transfer', None), ('primordial', None)]:
                    if nb * nk * (30 if z is not None else 1) > 2**30:
                        continue
                    power.analytic(engine, what, k, z=z, bg=bg, pk=pk, device=dev)
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    for _ in range(5):
                        power.analytic(engine, what, k, z=z, bg=bg, pk=pk, device=dev)
                    torch.cuda.synchronize()
                    dt = (time.perf_counter() - t0) / 5
                    print('%-24s ncosmo %6d nk %6d %-10s nz %2d: %8.1f us  %6.2f G (cosmo, k)/s' % (engine, nb, nk, what, 0 if z is None else len(z), dt * 1e6, nb * nk / dt / 1e9))


if __name__ == '__main__':
    main()
