"""cProfile of the host side of one wallish2018 / brieden2022 chunk (config 4).   python tools/host_profile_config4.py wallish2018"""
import cProfile
import os
import pstats
import sys
import warnings

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch      # noqa: E402
import bench      # noqa: E402
import cosmoprimo_amd as cp      # noqa: E402
from cosmoprimo_amd.bao_filter import PowerSpectrumBAOFilter      # noqa: E402

engine = sys.argv[1] if len(sys.argv) > 1 else 'wallish2018'
dev = torch.device('cuda', 0)
chunk, nchunks = 16384, 8
par = bench.eh_parameters(nchunks * chunk, 2, torch, dev)
warnings.simplefilter('ignore')
fid = cp.Cosmology(engine='eisenstein_hu')
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
prof = cProfile.Profile()
prof.enable()
for i in range(nchunks):
    run(slice(i * chunk, (i + 1) * chunk))
prof.disable()
torch.cuda.synchronize()
st = pstats.Stats(prof)
st.sort_stats('cumulative').print_stats(45)
if len(sys.argv) > 2:      # who calls the small tensor operations
    import io
    buf = io.StringIO()
    st = pstats.Stats(prof, stream=buf)
    st.print_callers('TensorBase|_tensor.py|torch._C._VariableFunctions')
    lines = [l for l in buf.getvalue().splitlines() if 'cosmoprimo_amd' in l or 'was called by' in l or '<-' in l or 'TensorBase' in l or '_VariableFunctions' in l]
    print('\n'.join(lines[:150]))
