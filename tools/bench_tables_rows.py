"""cp_tables_rows alone: 10 000 tables of 30 x 504 -> 64 x 1024, with and without the 10^x epilogue (post_op), against the two separate kernels.
    python tools/bench_tables_rows.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch      # noqa: E402
from cosmoprimo_amd import _lib, _device as dv      # noqa: E402
from cosmoprimo_amd.spline import LinearOperator, dense_operator      # noqa: E402

dev = torch.device('cuda', 0)
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
x, xq = np.linspace(-7., 2., 504), np.linspace(-7., 2., 1024)
zk, zq = np.linspace(0., 3., 30), np.linspace(0., 3., 64)
opx = LinearOperator.spline(x, xq, bc='not-a-knot', extrapolate=True, device=dev)
opz = LinearOperator.dense(dense_operator(zk, zq, bc='not-a-knot', extrapolate=True), device=dev)
t = torch.rand((nb, 30, 504), dtype=torch.float64, device=dev)
out = torch.empty((nb, 64, 1024), dtype=torch.float64, device=dev)
lib = _lib.load()


def timeit(fn, label):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 10 * 1e3
    print('%-46s %.3f ms  (%.2f TB/s of input + output)' % (label, ms, (t.numel() + out.numel()) * 8 / ms / 1e9))


for post, name in ((2, '10^x'), (0, 'none'), (1, 'sqrt')):
    timeit(lambda: _lib.check(lib.cp_tables_rows(opx._handle, opz._handle, t.data_ptr(), out.data_ptr(), nb, post, 1., dv.stream_of(dev))), 'cp_tables_rows, epilogue ' + name)
timeit(lambda: opz.mid(opx(t), post='exp10'), 'k operator, then middle-axis GEMM with 10^x')
timeit(lambda: opx(t), '  of which the k operator')
