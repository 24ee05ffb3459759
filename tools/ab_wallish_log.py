"""A/B in one process: wallish2018 with log(k P) from the evaluation kernel vs the logarithm taken in the transform.   python tools/ab_wallish_log.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch      # noqa: E402
import bench      # noqa: E402
import cosmoprimo_amd as cp      # noqa: E402
from cosmoprimo_amd import bao_filter      # noqa: E402

dev = torch.device('cuda', 0)
params = bench.eh_parameters(4 * 16384, 2, torch, dev)
orig = bao_filter.Wallish2018PowerSpectrumBAOFilter._log_k_rows
for rep in range(3):
    for name, fn in (('log in evaluation', orig), ('log in transform', lambda self, k: None)):
        bao_filter.Wallish2018PowerSpectrumBAOFilter._log_k_rows = fn
        out = bench.config4(cp, torch, dev, params, engines=('wallish2018',))
        print(name, {n: '%.3g vectors/s, %.3f ms per chunk' % (v['value'], v['ms'] / 4) for n, v in out.items()})
