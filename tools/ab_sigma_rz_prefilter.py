"""config 3 (sigma_rz 256 r x 64 z of 10 000 EH98 cosmologies, bench.py's call) with the fused kernel's spline from B-spline coefficients (round 6) against the
banded operator out of L2, alternately in one process; the largest difference between the two results.   python tools/ab_sigma_rz_prefilter.py"""
import os
import sys
import warnings

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch      # noqa: E402
import bench      # noqa: E402
import cosmoprimo_amd as cp      # noqa: E402
from cosmoprimo_amd import interpolator as itp      # noqa: E402

warnings.simplefilter('ignore')
dev = torch.device('cuda:0')
for rnd in range(3):
    for flag in (True, False):
        itp._SIGMA_RZ_PREFILTERED = flag
        r3 = bench.config3(cp, torch, dev, reps=20)
        print('%-28s config 3 %.4f ms' % ('B-spline coefficients' if flag else 'banded operator', r3['ms']))
