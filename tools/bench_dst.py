"""The two DST launches of a wallish2018 chunk: 16 384 rows of 4096, fused log(k x) forward, fused exp(.)/k inverse, split coefficient layout.
    python tools/bench_dst.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch      # noqa: E402
from cosmoprimo_amd.dst import DST      # noqa: E402

dev = torch.device('cuda', 0)
n, nrows = 4096, 16384
klin = np.linspace(1e-7, 2., n)
d = DST(n, kx=klin, device=dev)
x = torch.rand((nrows, n), dtype=torch.float64, device=dev) + 0.5
y = d(x, fused=True, split=True)
for label, fn in (('forward, fused log', lambda: d(x, fused=True, split=True)), ('inverse, fused exp / k', lambda: d(y, inverse=True, fused=True, split=True)),
                  ('forward, plain', lambda: d(x)), ('inverse, plain', lambda: d(y, inverse=True))):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 20 * 1e3
    print('%-24s %.3f ms  (%.2f TB/s)' % (label, ms, 2 * nrows * n * 8 / ms / 1e9))
back = d(y, inverse=True, fused=True, split=True)
print('round trip error %.2e' % float((back / x - 1).abs().max()))
