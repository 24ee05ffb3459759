"""The FFTLog fuzz fixtures on the GPU: error of every configuration next to the movement of the reference's own result under one rounding error per padded,
prefactored input sample (recorded in the fixture: `moves`) -- the numbers the factor in tests/test_fftlog_fuzz_gpu.py / test_fftlog_large_gpu.py comes from.
    python tools/fuzz_error_distribution.py > profiles/r6_fftlog_fuzz_errors.txt"""
import os
import sys

import numpy as np

sys.path.insert(0, os.getcwd())
from oracle.gen_golden import fftlog_fuzz_configs, fftlog_fuzz_build, fftlog_fuzz_stride, fftlog_fuzz_error, fftlog_large_configs
from cosmoprimo_amd import fftlog as fl

GOLDEN = os.path.join(os.getcwd(), 'tests', 'golden')
for name, configs in (('fftlog_fuzz', fftlog_fuzz_configs()), ('fftlog_large', fftlog_large_configs())):
    g = np.load(os.path.join(GOLDEN, name + '.npz'))
    ratios = []
    print('== %s: config, error (tilted, norm-wise), moves of the reference, error / moves' % name)
    for i, cfg in enumerate(configs):
        obj, x, fun = fftlog_fuzz_build(fl, cfg)
        y, out = obj(fun, extrap=cfg['extrap'], keep_padding=cfg['keep_padding'])
        y, out = np.asarray(y), np.asarray(out)
        stride = fftlog_fuzz_stride(y.shape[-1])
        err = fftlog_fuzz_error(cfg, out[..., ::stride], g['c%d_g' % i], g['c%d_y' % i])
        moves = float(g['c%d_moves' % i])
        ratio = err / moves if moves > 0 else float('nan')
        flag = '' if err <= 1e-12 else '   <- above 1e-12: held to the movement'
        if err > 1e-12:
            ratios.append(ratio)
        print('%3d  err %.3e  moves %.3e  err/moves %8.3f%s' % (i, err, moves, ratio, flag))
    if ratios:
        print('-- configurations above 1e-12: %d; error / moves: max %.3f, median %.3f' % (len(ratios), max(ratios), float(np.median(ratios))))
    else:
        print('-- no configuration above 1e-12')
