"""One real row per workgroup (cp_rfft_forward / backward: an N/2-point complex FFT + a split pass, 16 N bytes moved per row) against the DST of the same rows (two rows
per workgroup as one N-point complex FFT): what a transform with half the LDS per workgroup reaches.   python tools/bench_rfft_rows.py [nrows] [n]"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch      # noqa: E402
from cosmoprimo_amd import _lib, _device as dv      # noqa: E402
from cosmoprimo_amd.dst import DST      # noqa: E402

nrows = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
dev = torch.device('cuda', 0)
lib, st = _lib.load(), dv.stream_of(dev)
x = torch.rand((nrows, n), dtype=torch.float64, device=dev)
spec = torch.empty((nrows, n // 2 + 1, 2), dtype=torch.float64, device=dev)
plan = ctypes.c_void_p()
_lib.check(lib.cp_rfft_plan_create(ctypes.byref(plan), n, 0))


def timed(fun, label, nbytes):
    for _ in range(3):
        fun()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        fun()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print('%-44s %.3f ms  %.2f TB/s' % (label, ms, nbytes / ms / 1e9))


timed(lambda: _lib.check(lib.cp_rfft_forward(plan, x.data_ptr(), spec.data_ptr(), nrows, st)), 'rfft forward, a row per workgroup', 16. * n * nrows)
timed(lambda: _lib.check(lib.cp_rfft_backward(plan, spec.data_ptr(), x.data_ptr(), nrows, 0, st)), 'rfft backward, a row per workgroup', 16. * n * nrows)
dst = DST(n, device=dev)
y = torch.empty_like(x)
h = dst._handle
timed(lambda: _lib.check(lib.cp_dst_execute(h, x.data_ptr(), y.data_ptr(), nrows, 0, 0, st)), 'DST-II, two rows per workgroup', 16. * n * nrows)
timed(lambda: _lib.check(lib.cp_dst_execute(h, x.data_ptr(), y.data_ptr(), nrows, 1, 0, st)), 'DST-III (inverse), two rows per workgroup', 16. * n * nrows)
