"""Probe (development aid): can one chunk of config 4 (Cosmology -> P(k) -> filter) be captured into a HIP graph and replayed?  python tools/graph_probe.py [engine]"""
import os
import sys
import time
import warnings

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    import bench
    import cosmoprimo_amd as cp
    from cosmoprimo_amd.bao_filter import PowerSpectrumBAOFilter
    engine = sys.argv[1] if len(sys.argv) > 1 else 'wallish2018'
    dev = torch.device('cuda', 0)
    torch.cuda.set_device(dev)
    n = int(sys.argv[2]) if len(sys.argv) > 2 else bench.CONFIG4_CHUNK
    par = bench.eh_parameters(2 * n, 2, torch, dev)
    static = {name: v[:n].clone() for name, v in par.items()}
    warnings.simplefilter('ignore')
    fid = cp.Cosmology(engine='eisenstein_hu')
    state = {}

    def chunk():
        cosmo = cp.Cosmology(engine='eisenstein_hu', sigma8=0.8, **static)
        interp = cosmo.get_fourier().pk_interpolator(z=np.array([0.]))
        kw = dict(cosmo_fid=fid, cosmo=cosmo) if engine == 'brieden2022' else {}
        if 'filter' not in state:
            state['filter'] = PowerSpectrumBAOFilter(interp, engine=engine, **kw)
        else:
            state['filter'](interp, cosmo=cosmo if kw else None)
        return state['filter']._pknow_rows

    for _ in range(3):
        ref = chunk().clone()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        chunk()
    torch.cuda.synchronize()
    print('eager: %.3f ms per chunk' % ((time.perf_counter() - t0) / 20 * 1e3))
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream(dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):
        chunk()
    torch.cuda.current_stream(dev).wait_stream(side)
    with torch.cuda.graph(g):
        out = chunk()
    g.replay()
    torch.cuda.synchronize()
    print('replay equals eager:', bool(torch.equal(out, ref)))
    for name in static:
        static[name].copy_(par[name][n:])
    g.replay()
    torch.cuda.synchronize()
    second = out.clone()
    eager2 = chunk()
    print('second block equals eager:', bool(torch.equal(second, eager2)), float((second / eager2 - 1).abs().max()))
    t0 = time.perf_counter()
    for _ in range(20):
        g.replay()
    torch.cuda.synchronize()
    print('graph replay: %.3f ms per chunk' % ((time.perf_counter() - t0) / 20 * 1e3))


if __name__ == '__main__':
    main()
