"""config 3's call (sigma_rz of 10 000 EH98 cosmologies, 256 r x 64 z) timed WITHOUT the parity check of bench.config3 -- for the diagnostic builds whose
results are wrong on purpose (tools/sigma_ablate_spline.sh).   python tools/time_config3_raw.py"""
import os
import sys
import time
import warnings

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch      # noqa: E402
import bench      # noqa: E402
import cosmoprimo_amd as cp      # noqa: E402

warnings.simplefilter('ignore')
dev = torch.device('cuda', 0)
cosmo = cp.Cosmology(engine='eisenstein_hu', sigma8=0.8, **bench.eh_parameters(10000, 1, torch, dev))
interp = cosmo.get_fourier().pk_interpolator()
r, z = torch.as_tensor(np.geomspace(1, 100, 256), device=dev), torch.as_tensor(np.linspace(0, 3, 64), device=dev)
for rnd in range(3):
    for _ in range(5):
        interp.sigma_rz(r, z)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        interp.sigma_rz(r, z)
    torch.cuda.synchronize()
    print('config 3 call: %.4f ms' % ((time.perf_counter() - t0) / 20 * 1e3))
