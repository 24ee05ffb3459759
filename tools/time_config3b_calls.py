import os, sys, time
sys.path.insert(0, '.')
import numpy as np, torch
import cosmoprimo_amd as cp
g = np.load('tests/golden/sigma.npz')
rng = np.random.default_rng(1)
nb = 10000
amp = torch.as_tensor(rng.uniform(0.5, 2., nb), device='cuda')
batch = amp[:, None, None] * torch.as_tensor(g['table_pk'], device='cuda')[None]
r, zq = torch.as_tensor(g['r'], device='cuda'), torch.as_tensor(g['z'], device='cuda')
interp = cp.PowerSpectrumInterpolator2D(g['table_k'], g['table_z'], batch)
import cosmoprimo_amd.interpolator as it
for flag in (True, False, True, False):
  it._TRANSPOSE_IN_STORE = flag
  print('transposition in the store' if flag else 'separate copy', end=': ')
  for i in range(8):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = interp.sigma_rz(r, zq).contiguous()
    torch.cuda.synchronize(); print('%.2f' % ((time.perf_counter() - t0) * 1e3), end=' ')
  print()
print(torch.cuda.memory_reserved() / 1e9, 'GB reserved')
