"""config 4 on one GPU's share (125 000 vectors) for several chunk sizes: vectors/s per filter.   python tools/chunk_config4.py [chunk ...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch      # noqa: E402
import bench      # noqa: E402
import cosmoprimo_amd as cp      # noqa: E402

dev = torch.device('cuda', 0)
par = bench.eh_parameters(125000, 2, torch, dev)
for chunk in [int(a) for a in sys.argv[1:]] or [16384, 32768, 65536, 125000]:
    for rep in range(2):
        out = bench.config4(cp, torch, dev, par, chunk=chunk, spot_check=False)
        print(chunk, {name: '%.3g vectors/s, %.2f ms' % (v['value'], v['ms']) for name, v in out.items()}, flush=True)
