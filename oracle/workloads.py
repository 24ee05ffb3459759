"""Synthetic workloads of BASELINE.json's configs; shared by tests and bench (TEST/BENCH INFRASTRUCTURE)."""
import os
import numpy as np

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests', 'golden')


def pk_eh_default(n):
    """EH1998 P(k, z=0) of the default cosmology on k=logspace(-5,2,n), n in {1024, 2048} (golden fixture from the reference)."""
    d = np.load(os.path.join(GOLDEN_DIR, 'pk_eh_default.npz'))
    return d['k%d' % n], d['pk%d' % n]


def config2_rows(k, pk_base, start, stop, seed=0, nbatch=100000):
    """Rows [start, stop) of the config-2 batch: pk_b = A_b (k/0.05)^dn_b P_EH(k), A~U(0.5,2), dn~U(-0.1,0.1), default_rng(seed)."""
    rng = np.random.default_rng(seed)
    amp = rng.uniform(0.5, 2., nbatch)
    dn = rng.uniform(-0.1, 0.1, nbatch)
    return amp[start:stop, None] * (k[None, :] / 0.05) ** dn[start:stop, None] * pk_base[None, :]
