import os, sys, time, warnings, cProfile, pstats
import numpy as np
sys.path.insert(0, os.getcwd())
import torch
import cosmoprimo_amd as cp
warnings.simplefilter('ignore')
nb = 10000
dev = torch.device('cuda', 0)
rng = np.random.default_rng(1)
par = dict(Omega_m=rng.uniform(.25, .40, nb), Omega_b=rng.uniform(.04, .06, nb), h=rng.uniform(.6, .8, nb), n_s=rng.uniform(.92, 1., nb), sigma8=0.8)
r, z = np.geomspace(1, 100, 256), np.linspace(0, 3, 64)
cosmo = cp.Cosmology(engine='eisenstein_hu', **{k: (torch.as_tensor(v, device=dev) if np.ndim(v) else v) for k, v in par.items()})
interp = cosmo.get_fourier().pk_interpolator()
rt, zt = torch.as_tensor(r, device=dev), torch.as_tensor(z, device=dev)
for i in range(3):
    interp.sigma_rz(rt, zt)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(20):
    out = interp.sigma_rz(rt, zt)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print('host enqueue %.3f ms per call, total %.3f ms per call' % ((t1 - t0) / 20 * 1e3, (t2 - t0) / 20 * 1e3))
pr = cProfile.Profile()
pr.enable()
for i in range(20):
    out = interp.sigma_rz(rt, zt)
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(35)
