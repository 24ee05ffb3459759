"""cp_wallish_dd_box on 32 768 sequences of 2048 coefficients (one chunk of wallish2018 in config 4): second derivatives + boxes, with and without the
removal of the boxes in place.    python tools/bench_dd_box.py [nseq]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    from cosmoprimo_amd import _lib, _device as dv
    nseq, n = int(sys.argv[1]) if len(sys.argv) > 1 else 32768, 2048
    dev = torch.device('cuda', 0)
    gen = torch.Generator(device=dev).manual_seed(1)
    x = 1. + torch.arange(n, device=dev, dtype=torch.float64)
    y = torch.randn((nseq, n), generator=gen, device=dev, dtype=torch.float64) / x**1.5 + 3e-3 * torch.exp(-0.5 * ((x - 0.35 * n) / 12.)**2)
    box = torch.empty((nseq, 2), dtype=torch.int32, device=dev)
    work = y.clone()
    lib = _lib.load()
    for label, gap in (('second derivatives + boxes', None), ('... + removal of the boxes in place', work)):
        ms = []
        for rep in range(3):
            if gap is not None:
                work.copy_(y)
            for _ in range(2):
                _lib.check(lib.cp_wallish_dd_box(y.data_ptr(), nseq, n, 20, 5, -10, 20, box.data_ptr(), None, gap.data_ptr() if gap is not None else None, 0, dv.stream_of(dev)))
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                _lib.check(lib.cp_wallish_dd_box(y.data_ptr(), nseq, n, 20, 5, -10, 20, box.data_ptr(), None, gap.data_ptr() if gap is not None else None, 0, dv.stream_of(dev)))
            e1.record()
            torch.cuda.synchronize()
            ms.append(e0.elapsed_time(e1) / 10)
        print('%-40s %s ms per %d sequences' % (label, ' '.join('%.3f' % v for v in ms), nseq))


if __name__ == '__main__':
    main()
