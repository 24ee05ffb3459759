"""The inverse DST of wallish2018 (exp(idst(.)) / k_lin, split layout) on 32 768 rows of 4096 coefficients.    python tools/bench_dst.py [nrows]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    from cosmoprimo_amd.dst import DST
    nrows, n = int(sys.argv[1]) if len(sys.argv) > 1 else 32768, 4096
    dev = torch.device('cuda', 0)
    klin = np.linspace(1e-7, 2., n)
    dst = DST(n, kx=klin, device=dev)
    gen = torch.Generator(device=dev).manual_seed(1)
    y = torch.randn((nrows, n), generator=gen, device=dev, dtype=torch.float64) * 0.1
    for label, kw in (('inverse, fused exp / k, split', dict(inverse=True, fused=True, split=True)), ('inverse, plain', dict(inverse=True)), ('forward, plain', dict())):
        ms = []
        for rep in range(3):
            for _ in range(2):
                out = dst(y, **kw)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                out = dst(y, **kw)
            e1.record()
            torch.cuda.synchronize()
            ms.append(e0.elapsed_time(e1) / 10)
        print('%-34s %s ms per %d rows' % (label, ' '.join('%.3f' % v for v in ms), nrows))


if __name__ == '__main__':
    main()
