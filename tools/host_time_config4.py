"""config 4: how long the host takes to QUEUE a chunk of each filter against how long the device takes to run it (is the host ahead?).
python tools/host_time_config4.py"""
import os
import sys
import time
import warnings

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch      # noqa: E402
import bench      # noqa: E402
import cosmoprimo_amd as cp      # noqa: E402
from cosmoprimo_amd.bao_filter import PowerSpectrumBAOFilter      # noqa: E402

dev = torch.device('cuda', 0)
chunk, nchunks = 16384, 8
par = bench.eh_parameters(nchunks * chunk, 2, torch, dev)
warnings.simplefilter('ignore')
fid = cp.Cosmology(engine='eisenstein_hu')
for engine in ('wallish2018', 'brieden2022'):
    kw = dict(cosmo_fid=fid) if engine == 'brieden2022' else {}
    state = {}

    def run(sl):
        cosmo = cp.Cosmology(engine='eisenstein_hu', sigma8=0.8, **{name: v[sl] for name, v in par.items()})
        interp = cosmo.get_fourier().pk_interpolator(z=np.array([0.]))
        if 'filter' not in state:
            state['filter'] = PowerSpectrumBAOFilter(interp, engine=engine, **(dict(kw, cosmo=cosmo) if kw else {}))
        else:
            state['filter'](interp, cosmo=cosmo if kw else None)
        return state['filter']._pknow_rows

    for _ in range(6):
        run(slice(0, chunk))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(nchunks):
        run(slice(i * chunk, (i + 1) * chunk))
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print('%s: host queues a chunk in %.2f ms; device done %.2f ms per chunk (%.3g vectors/s)' % (engine, (t1 - t0) / nchunks * 1e3, (t2 - t0) / nchunks * 1e3,
                                                                                               nchunks * chunk / (t2 - t0)))
