"""Probe (development aid): do the ALU-bound P(k) kernel and the write-bound sigma(r, z) store kernel of config 3 overlap when they are issued on two
streams for different blocks of cosmologies?   python tools/overlap_probe.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    from cosmoprimo_amd import power
    from cosmoprimo_amd.spline import LinearOperator
    dev = torch.device('cuda', 0)
    rng = np.random.default_rng(0)
    nb = 5000
    bg = dict(Omega_cdm=torch.as_tensor(rng.uniform(.2, .35, nb), device=dev), Omega_b=torch.as_tensor(rng.uniform(.04, .06, nb), device=dev),
              h=torch.as_tensor(rng.uniform(.6, .8, nb), device=dev))
    pk = dict(A_s=2e-9, n_s=torch.as_tensor(rng.uniform(.92, 1., nb), device=dev))
    k = torch.as_tensor(np.geomspace(1e-7, 1e2, 1024), device=dev)
    x = np.geomspace(1e-2, 1e2, 1024)
    op = LinearOperator.spline(np.log(x), np.log(np.geomspace(1., 100., 256)), bc='natural', device=dev)
    var = torch.as_tensor(rng.uniform(0.5, 2., size=(nb, 1024)) * x**-0.7, device=dev)
    g = torch.as_tensor(rng.uniform(0.1, 1., size=(nb, 64)), device=dev)
    sa, sb = torch.cuda.Stream(dev), torch.cuda.Stream(dev)

    def run_a(n):
        with torch.cuda.stream(sa):
            for _ in range(n):
                power.analytic('eisenstein_hu', 'matter', k, bg=bg, pk=pk, device=dev)

    def run_b(n):
        with torch.cuda.stream(sb):
            for _ in range(n):
                op.outer(var, g, sqrt=True)

    def timed(fn):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) * 1e3

    n = 40
    ta, tb = timed(lambda: run_a(n)), timed(lambda: run_b(n))

    def both():
        for _ in range(n):
            run_a(1)
            run_b(1)
    tab = timed(both)
    print('P(k) alone %.3f ms, store kernel alone %.3f ms per block of %d; interleaved on two streams %.3f ms (sum %.3f, max %.3f)' % (
        ta / n, tb / n, nb, tab / n, (ta + tb) / n, max(ta, tb) / n))


if __name__ == '__main__':
    main()
