"""FFTLog + spline to 256 radii: the fused kernel (cp_fftlog_spline_execute) against the two separate kernels, by number of rows.
    python tools/bench_fused_spline.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch      # noqa: E402
import cosmoprimo_amd.interpolator as it      # noqa: E402

dev = torch.device('cuda', 0)
r = np.geomspace(1., 100., 256)
for nrows in (64, 512, 4096, 32768, 262144):
    rows = torch.rand((nrows, 1024), dtype=torch.float64, device=dev) + 0.5
    line = '%7d rows:' % nrows
    for label, window in (('fused', (2, 1 << 40)), ('separate', (0, -1))):
        it._FUSED_SPLINE_ROWS = window
        fn = lambda: it.sigma_r2_of_rows(r, lambda k: rows, device=dev, sqrt=True)      # noqa: E731
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            fn()
        torch.cuda.synchronize()
        line += '  %s %.3f ms' % (label, (time.perf_counter() - t0) / 20 * 1e3)
    print(line)
