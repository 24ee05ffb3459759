"""Config 3B (tools/bench_config3b.py) with and without the windowed stores of the FFTLog (cp_fftlog_execute_window), alternately in one process on
one box (boxes differ by 3 %).   python tools/ab_fftlog_window.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch      # noqa: E402
import cosmoprimo_amd as cp      # noqa: E402
from cosmoprimo_amd.spline import LinearOperator      # noqa: E402

g = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden', 'sigma.npz'))
rng = np.random.default_rng(1)
nb = 10000
batch = torch.as_tensor(rng.uniform(0.5, 2., nb), device='cuda')[:, None, None] * torch.as_tensor(g['table_pk'], device='cuda')[None]
r, zq = torch.as_tensor(g['r'], device='cuda'), torch.as_tensor(g['z'], device='cuda')
interp = cp.PowerSpectrumInterpolator2D(g['table_k'], g['table_z'], batch)
windowed = LinearOperator.columns
whole = property(lambda self: (0, self.n))


def run():
    for _ in range(4):
        interp.sigma_rz(r, zq)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        interp.sigma_rz(r, zq)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / 10 * 1e3


for rep in range(4):
    LinearOperator.columns = whole
    a = run()
    LinearOperator.columns = windowed
    b = run()
    print('whole rows %.3f ms   window %.3f ms' % (a, b))
