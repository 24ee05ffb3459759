"""Config 3B with the rows of P(k, z) kept on the die between the two kernels: the batch of tables walked in chunks whose rows (512 KB per table)
fit the 256 MiB last-level cache, the SAME scratch buffer written by cp_tables_rows_direct and read by cp_fftlog_geospline_execute for every chunk.
The two library calls of one ordinary sigma_rz are recorded and replayed with shifted pointers.
    python tools/chunk_config3b.py [ntables] [chunk sizes ...]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    import cosmoprimo_amd as cp
    from cosmoprimo_amd import _lib
    nb = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
    chunks = [int(v) for v in sys.argv[2:]] or [nb, 1536, 768, 384, 192, 96]
    g = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden', 'sigma.npz'))
    rng = np.random.default_rng(1)
    amp = torch.as_tensor(rng.uniform(0.5, 2., nb), device='cuda')
    batch = amp[:, None, None] * torch.as_tensor(g['table_pk'], device='cuda')[None]
    r, zq = torch.as_tensor(g['r'], device='cuda'), torch.as_tensor(g['z'], device='cuda')
    interp = cp.PowerSpectrumInterpolator2D(g['table_k'], g['table_z'], batch)
    ref = interp.sigma_rz(r, zq)
    lib = _lib.load()
    seen = {}
    real = {name: getattr(lib, name) for name in ('cp_tables_rows_direct', 'cp_fftlog_geospline_execute')}

    def recorder(name):
        def call(*args):
            seen[name] = args
            return real[name](*args)
        return call

    for name in real:
        setattr(lib, name, recorder(name))
    ref2 = interp.sigma_rz(r, zq)
    for name in real:
        setattr(lib, name, real[name])
    assert torch.equal(ref, ref2)
    ta, ga = seen['cp_tables_rows_direct'], seen['cp_fftlog_geospline_execute']
    kplan, opy, t_ptr, m_ptr, rows_ptr, ntab, post, scale, stream = ta
    fplan, splan, rows_ptr2, out_ptr, nrows, group, sq, stream2 = ga
    assert ntab == nb and rows_ptr == rows_ptr2 and nrows % nb == 0 and group == nrows // nb
    nzq = group
    nz, nx = interp._interp._fun_y_major.shape[-2:]
    nk = 1024
    nr = ref.shape[-2]
    t_stride, row_stride, out_stride = nz * nx * 8, nzq * nk * 8, nr * nzq * 8
    print('tables %d x (%d z, %d k) -> rows %d x %d x %d -> (%d r, %d z)' % (nb, nz, nx, nb, nzq, nk, nr, nzq))
    tables, direct = real['cp_tables_rows_direct'], real['cp_fftlog_geospline_execute']
    for chunk in chunks:
        scratch = torch.empty((min(chunk, nb), nzq, nk), dtype=torch.float64, device='cuda')
        out = torch.zeros_like(ref)

        def walk():
            for t0 in range(0, nb, chunk):
                nc = min(chunk, nb - t0)
                _lib.check(tables(kplan, opy, t_ptr + t0 * t_stride, m_ptr + t0 * t_stride, scratch.data_ptr(), nc, post, scale, stream))
                _lib.check(direct(fplan, splan, scratch.data_ptr(), out.data_ptr() + t0 * out_stride, nc * nzq, group, sq, stream))

        for _ in range(3):
            walk()
        torch.cuda.synchronize()
        ms = []
        for rep in range(4):
            t0 = time.perf_counter()
            for _ in range(5):
                walk()
            torch.cuda.synchronize()
            ms.append((time.perf_counter() - t0) / 5 * 1e3)
        assert torch.equal(out, ref), chunk
        print('chunks of %6d tables (%7.1f MB of rows)   %s ms' % (chunk, min(chunk, nb) * row_stride / 1e6, ' '.join('%.3f' % v for v in ms)))


if __name__ == '__main__':
    main()
