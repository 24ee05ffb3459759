"""640 000 rows of 1024 samples (the FFTLog stage of config 3B) -> 256 radii, alternately in one process: cp_fftlog_execute_window + the band
operator on the matrix cores (round 3) against cp_fftlog_geospline_execute (the spline solved on the CU, round 4, or folded into the transform, round 6),
plain and grouped layouts.
    python tools/bench_geospline.py [nrows]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    import cosmoprimo_amd as cp
    from cosmoprimo_amd import interpolator as itp
    from cosmoprimo_amd.spline import LinearOperator
    nrows = int(sys.argv[1]) if len(sys.argv) > 1 else 640000
    dev = torch.device('cuda', 0)
    k = np.geomspace(1e-7, 1e2, 1024)
    fft = cp.TophatVariance(k, device=dev)
    s, r = fft.y[0], np.geomspace(1., 100., 256)
    op = LinearOperator.spline(s, r, bc='natural', device=dev)
    rng = np.random.default_rng(0)
    base = torch.as_tensor((k / 0.05)**-1.5 * 1e3 / (1. + (k / 0.02)**2.2), device=dev)
    rows = (torch.as_tensor(rng.uniform(0.5, 2., (nrows, 1)), device=dev) * base[None, :]).contiguous()
    def geo(prefiltered, shaped, group):      # the spline's solve folded into the transform (round 6) or run on the CU (round 4)
        itp._GEOSPLINE_PREFILTERED = prefiltered
        return itp._fftlog_then_geospline(fft, s, r, shaped, dev, sqrt=True, group=group)

    routes = {
        'fftlog (windowed stores) + operator': lambda: op(fft(rows, out_window=op.columns)[1], sqrt=True),
        'fftlog + operator, grouped store': lambda: op(fft(rows.reshape(-1, 64, 1024), out_window=op.columns)[1], sqrt=True, last_axis_first=True),
        'geospline, (rows, radii)': lambda: geo(True, rows, 0),
        'geospline, (tables, radii, 64)': lambda: geo(True, rows.reshape(-1, 64, 1024), 64),
        'geospline solved on the CU, (rows, radii)': lambda: geo(False, rows, 0),
        'geospline solved on the CU, (tables, radii, 64)': lambda: geo(False, rows.reshape(-1, 64, 1024), 64),
    }
    res = {}
    for rnd in range(3):
        for name, fn in routes.items():
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(10):
                fn()
            torch.cuda.synchronize()
            res.setdefault(name, []).append((time.perf_counter() - t0) / 10 * 1e3)
    for name, ms in res.items():
        print('%-50s %s ms' % (name, ' '.join('%.3f' % v for v in ms)))
    a, b = (geo(p, rows[:4096], 0) for p in (True, False))
    print('prefiltered against solved, 4096 rows: max relative difference %.3g' % float(((a - b).abs() / b.abs()).max()))


if __name__ == '__main__':
    main()
