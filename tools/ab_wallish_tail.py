"""wallish2018 on bench.py's config-4 share with and without cp_wallish_tail (everything behind the forward transform as one kernel), alternately in one
process.   python tools/ab_wallish_tail.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch      # noqa: E402
import bench      # noqa: E402
import cosmoprimo_amd as cp      # noqa: E402
from cosmoprimo_amd import bao_filter as bf      # noqa: E402

dev = torch.device('cuda', 0)
par = bench.eh_parameters(125000, 2, torch, dev)
for rep in range(3):
    for label, fused, whole in (('whole filter in one kernel', True, True), ('tail in one kernel', True, False), ('three kernels', False, False)):
        bf._TAIL_IN_ONE_KERNEL, bf._ALL_IN_ONE_KERNEL = fused, whole
        out = bench.config4(cp, torch, dev, par, engines=('wallish2018',), spot_check=False)
        print(label, {name: '%.4g vectors/s, %.2f ms for the share' % (v['value'], v['ms']) for name, v in out.items()}, flush=True)
