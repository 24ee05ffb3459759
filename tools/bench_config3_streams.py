"""config 3 (sigma_rz of 10 000 EH98 cosmologies, 256 r x 64 z): the fused kernel (cp_sigma_rz_analytic, nblocks = 0), the two-stream block walk of
the three separate kernels with 1, 2, 4, 8 blocks, and the three separate calls.   python tools/bench_config3_streams.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch      # noqa: E402
import bench      # noqa: E402
import cosmoprimo_amd as cp      # noqa: E402

dev = torch.device('cuda', 0)
kind = cp.PowerSpectrumInterpolator2D
import os
ONLY = os.environ.get("CONFIG3_ONLY")
for label, blocks in (('fused kernel', 0), ('one call, 1 block', 1), ('one call, 2 blocks', 2), ('one call, 4 blocks', 4), ('one call, 8 blocks', 8), ('separate calls', None)):
    if blocks is None:
        kind._two_stream_min_bytes = 1 << 60
    else:
        kind._two_stream_blocks = blocks
    for rep in range(2):
        r = bench.config3(cp, torch, dev)
        print('%-20s ms %.4f events %.4f frac %.4f' % (label, r['ms'], r['ms_gpu_events'], r['roofline']['frac']))
