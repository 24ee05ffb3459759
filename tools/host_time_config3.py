"""configs 3 and 3B: time for the host to QUEUE one sigma_rz call against the time the device needs (a host synchronisation inside the call
shows as queue time = device time).   python tools/host_time_config3.py"""
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


def report(name, fn, reps=10):
    for _ in range(6):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print('%-28s host queues a call in %.3f ms; device done %.3f ms per call' % (name, (t1 - t0) / reps * 1e3, (t2 - t0) / reps * 1e3))


cosmo = cp.Cosmology(engine='eisenstein_hu', sigma8=0.8, **bench.eh_parameters(10000, 1, torch, dev))
interp = cosmo.get_fourier().pk_interpolator()
r, z = torch.as_tensor(np.geomspace(1, 100, 256), device=dev), torch.as_tensor(np.linspace(0, 3, 64), device=dev)
report('config 3 (analytic)', lambda: interp.sigma_rz(r, z))
g = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden', 'sigma.npz'))
rng = np.random.default_rng(1)
amp = torch.as_tensor(rng.uniform(0.5, 2., 10000), device=dev)
batch = amp[:, None, None] * torch.as_tensor(g['table_pk'], device=dev)[None]
tab = cp.PowerSpectrumInterpolator2D(g['table_k'], g['table_z'], batch)
rr, zq = torch.as_tensor(g['r'], device=dev), torch.as_tensor(g['z'], device=dev)
report('config 3B (tables)', lambda: tab.sigma_rz(rr, zq))
k = np.logspace(-4, 2, 2048)
f = cp.PowerToCorrelation(k, ell=0)
fun = torch.rand((100000, 2048), dtype=torch.float64, device=dev) + 0.5
report('FFTlog.__call__ 100000 x 2048', lambda: f(fun))
