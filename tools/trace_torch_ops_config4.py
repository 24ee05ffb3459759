"""Which lines of the package launch framework (torch) kernels during one steady-state chunk of config 4: every aten op dispatched while a chunk of
each filter runs, with the innermost frames inside cosmoprimo_amd/.   python tools/trace_torch_ops_config4.py [chunk]"""
import collections
import os
import sys
import traceback
import warnings

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    from torch.utils._python_dispatch import TorchDispatchMode
    import bench
    import cosmoprimo_amd as cp
    from cosmoprimo_amd.bao_filter import PowerSpectrumBAOFilter
    chunk = int(sys.argv[1]) if len(sys.argv) > 1 else bench.CONFIG4_CHUNK
    dev = torch.device('cuda', 0)
    par = bench.eh_parameters(2 * chunk, 2, torch, dev)
    warnings.simplefilter('ignore')
    fid = cp.Cosmology(engine='eisenstein_hu')

    class Log(TorchDispatchMode):
        def __init__(self):
            super().__init__()
            self.sites = collections.Counter()

        def __torch_dispatch__(self, func, types, args=(), kwargs=None):
            frames = [f for f in traceback.extract_stack() if 'cosmoprimo_amd' in f.filename]
            where = ' < '.join('%s:%d' % (os.path.basename(f.filename), f.lineno) for f in reversed(frames[-3:]))
            self.sites[(str(func), where)] += 1
            return func(*args, **(kwargs or {}))

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

        run(slice(0, chunk))
        run(slice(0, chunk))
        torch.cuda.synchronize()
        # calls into the library during the traced chunk (plan creations are synchronous allocations + uploads: none should be left per chunk)
        from cosmoprimo_amd import _lib
        lib = _lib.load()
        calls = collections.Counter()
        originals = {}
        for fname in _lib.SIGNATURES:
            fn = getattr(lib, fname)
            originals[fname] = fn

            def wrapped(*a, _fn=fn, _name=fname):
                calls[_name] += 1
                return _fn(*a)
            setattr(lib, fname, wrapped)
        with Log() as log:
            run(slice(chunk, 2 * chunk))
        for fname, fn in originals.items():
            setattr(lib, fname, fn)
        torch.cuda.synchronize()
        print('==== %s: library calls in one chunk: %s' % (engine, ', '.join('%s x%d' % kv for kv in sorted(calls.items()))))
        launching = {k: v for k, v in log.sites.items() if not any(s in k[0] for s in ('aten.view', 'aten.reshape', 'aten.slice', 'aten.select', 'aten.detach', 'aten.t.',
                                                                                      'aten.transpose', 'aten.expand', 'aten.unsqueeze', 'aten.squeeze', 'aten.alias',
                                                                                      'aten.empty', 'aten.as_strided', 'aten._unsafe_view', 'aten.permute', 'aten.is_', 'aten.lift'))}
        print('==== %s: %d aten ops that launch kernels in one chunk of %d (%d ops in all)' % (engine, sum(launching.values()), chunk, sum(log.sites.values())))
        for (op, where), n in sorted(launching.items(), key=lambda kv: kv[0][1]):
            print('%3d  %-34s %s' % (n, op.replace('aten.', ''), where))
        copies = {k: v for k, v in log.sites.items() if 'copy' in k[0] or 'zeros' in k[0] or 'scalar_tensor' in k[0] or 'full' in k[0]}
        print('     (of which copies / fills: %s)' % ', '.join('%s x%d @ %s' % (k[0].replace('aten.', ''), v, k[1]) for k, v in sorted(copies.items(), key=lambda kv: kv[0][1])))


if __name__ == '__main__':
    main()
