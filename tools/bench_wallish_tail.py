"""cp_wallish_tail on 65 536 vectors (a chunk of config 4), HIP events around 10 launches.   python tools/bench_wallish_tail.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np      # noqa: E402
import torch      # noqa: E402
import cosmoprimo_amd as cp      # noqa: E402
from cosmoprimo_amd import _lib, _device as dv      # noqa: E402
from cosmoprimo_amd.bao_filter import PowerSpectrumBAOFilter      # noqa: E402

dev = torch.device('cuda', 0)
rng = np.random.default_rng(2)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
par = dict(Omega_m=rng.uniform(0.24, 0.40, n), Omega_b=rng.uniform(0.04, 0.06, n), h=rng.uniform(0.6, 0.8, n), n_s=rng.uniform(0.92, 1., n))
cosmo = cp.Cosmology(engine='eisenstein_hu', sigma8=0.8, **par)
fid = cp.Cosmology(engine='eisenstein_hu')
interp = cosmo.get_fourier().pk_interpolator(z=np.array([0.]))
f = PowerSpectrumBAOFilter(interp, engine='wallish2018', cosmo=cosmo, cosmo_fid=fid)
ops = f._operators()
coef = f._even_now.new_empty((n, 4096))      # the rewritten coefficients as the filter left them: any smooth sequences serve the timing
coef.view(2 * n, 2048)[0::2] = f._even_now
coef.view(2 * n, 2048)[1::2] = f._odd_now
pk = f._pk_rows.contiguous()
box = torch.empty((2 * n, 2), dtype=torch.int32, device=dev)
out = torch.empty_like(pk)
lib = _lib.load()


def run():
    _lib.check(lib.cp_wallish_tail(ops['dst']._handle, ops['splice']._handle, coef.data_ptr(), pk.data_ptr(), pk.shape[1], n, 20, 5, -10, 20, ops['tophat'].data_ptr(),
                                   box.data_ptr(), out.data_ptr(), dv.stream_of(dev)))


for _ in range(5):
    run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
best = 1e9
for rep in range(3):
    e0.record()
    for _ in range(10):
        run()
    e1.record()
    torch.cuda.synchronize()
    best = min(best, e0.elapsed_time(e1) / 10)
print('cp_wallish_tail: %.3f ms per %d vectors' % (best, n))
