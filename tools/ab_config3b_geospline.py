"""Config 3B, alternately in one process: the round-3 route (FFTLog with windowed stores, then the band operator on the matrix cores) against the
FFTLog with the spline solved on the CU (cp_fftlog_geospline_execute), in sigma_rz's layout and in the plain (rows, radii) layout.
    python tools/ab_config3b_geospline.py [ncosmo]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    import cosmoprimo_amd as cp
    from cosmoprimo_amd import interpolator as itp
    nb = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
    g = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden', 'sigma.npz'))
    rng = np.random.default_rng(1)
    amp = torch.as_tensor(rng.uniform(0.5, 2., nb), device='cuda')
    batch = amp[:, None, None] * torch.as_tensor(g['table_pk'], device='cuda')[None]
    r, zq = torch.as_tensor(g['r'], device='cuda'), torch.as_tensor(g['z'], device='cuda')
    interp = cp.PowerSpectrumInterpolator2D(g['table_k'], g['table_z'], batch)
    modes = {'operator (round 3)': (1 << 60, True), 'solved, (r, z) layout': (8193, True), 'solved, (z, r) layout + view': (8193, False)}
    results = {}

    def run(mode, reps):
        itp._GEOSPLINE_MIN_ROWS, itp._GEOSPLINE_GROUPED = modes[mode]
        for _ in range(3):
            out = interp.sigma_rz(r, zq)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            out = interp.sigma_rz(r, zq)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps * 1e3, out

    ref = None
    for rnd in range(4):
        for mode in modes:
            ms, out = run(mode, 8)
            results.setdefault(mode, []).append(ms)
            if ref is None:
                ref = out.clone()
            else:
                err = float(((out - ref).abs() / ref.abs()).max())
                assert err < 1e-11, (mode, err)
    for mode, ms in results.items():
        print('%-32s %s ms' % (mode, ' '.join('%.3f' % v for v in ms)))


if __name__ == '__main__':
    main()
