"""config 4 as bench.py times it (65 536 EH98 vectors, four chunks per filter): vectors/s per filter.   python tools/bench_config4_quick.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch      # noqa: E402
import bench      # noqa: E402
import cosmoprimo_amd as cp      # noqa: E402

dev = torch.device('cuda', 0)
for rep in range(2):
    out = bench.config4(cp, torch, dev, bench.eh_parameters(4 * 16384, 2, torch, dev))
    print({name: '%.3g vectors/s, %.2f ms per chunk' % (v['value'], v['ms'] / 4) for name, v in out.items()})
