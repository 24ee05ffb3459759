"""CPU: the table interpolation of the 'tabulated' engine (csrc/cp_interp_table.h, compiled with g++: tests/host_emu/emu_interp.cpp) -- the law of the
knots the library finds (uniform, uniform in the logarithm behind leading knots as the reference's data/desi.dat, neither) and the kernel's per-sample
code (interval guessed from the law, walked to numpy's interval) against numpy.interp, bit for bit; also with a law the table does not follow.  The kernel
proper is tested on the GPU (tests/test_fiducial_gpu.py::test_interp_table_laws)."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT


def _lib():
    src = os.path.join(ROOT, 'tests', 'host_emu', 'emu_interp.cpp')
    out = os.path.join(ROOT, 'tests', 'host_emu', 'libemu_interp.so')
    dep = os.path.join(ROOT, 'cosmoprimo_amd', 'csrc', 'cp_interp_table.h')
    if not os.path.isfile(out) or os.path.getmtime(out) < max(os.path.getmtime(src), os.path.getmtime(dep)):
        subprocess.check_call(['g++', '-O1', '-std=c++17', '-ffp-contract=off', '-shared', '-fPIC', '-o', out, src])
    return ctypes.CDLL(out)


def as_p(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))


def find_law(x):
    law, first, a, b = ctypes.c_int(), ctypes.c_longlong(), ctypes.c_double(), ctypes.c_double()
    _lib().emu_interp_law(ctypes.c_longlong(x.size), as_p(x), ctypes.byref(law), ctypes.byref(first), ctypes.byref(a), ctypes.byref(b))
    return law.value, first.value, a.value, b.value


def apply(x, f, law, xq):
    out = np.empty_like(xq)
    outside = _lib().emu_interp_apply(ctypes.c_longlong(x.size), as_p(x), as_p(f), law[0], ctypes.c_longlong(law[1]), ctypes.c_double(law[2]), ctypes.c_double(law[3]),
                                      ctypes.c_longlong(xq.size), as_p(xq), as_p(out))
    return out, bool(outside)


def samples(x, rng, n=20000):
    q = [rng.uniform(x[0], x[-1], n), x, np.nextafter(x[1:], -np.inf), np.nextafter(x[:-1], np.inf), 0.5 * (x[1:] + x[:-1])]
    if x[0] > 0:
        q.append(np.exp(rng.uniform(np.log(x[0]), np.log(x[-1]), n)))
    return np.clip(np.concatenate(q), x[0], x[-1])


TABLES = {'uniform': (lambda rng: np.linspace(-2., 5., 3001), 1, 0),
          'desi': (lambda rng: np.concatenate([[0.], np.logspace(-8, 2, 40001)]), 2, 1),
          'geometric': (lambda rng: np.geomspace(3e-4, 7e5, 777), 2, 0),
          'three leading knots': (lambda rng: np.concatenate([[-1., 0., 1e-12], np.geomspace(1e-6, 10., 500)]), 2, 3),
          'nine leading knots': (lambda rng: np.concatenate([-np.arange(9., 0., -1.), np.geomspace(1e-6, 10., 500)]), 0, 0),
          'irregular': (lambda rng: np.sort(rng.uniform(0., 10., 5000)), 0, 0),
          'off by a third of a step': (lambda rng: np.linspace(0., 1., 200) + np.r_[0., 0.33 / 199 * np.sin(np.arange(1, 199)), 0.], 0, 0),
          'off by a fifth of a step': (lambda rng: np.linspace(0., 1., 200) + np.r_[0., 0.2 / 199 * np.sin(np.arange(1, 199)), 0.], 1, 0),
          'repeated knots': (lambda rng: np.repeat(np.linspace(0., 1., 50), 2), 0, 0),
          'three rows': (lambda rng: np.array([1., 2., 3.]), 1, 0),
          'two rows': (lambda rng: np.array([1., 2.]), 0, 0),
          'huge range': (lambda rng: np.geomspace(1e-300, 1e300, 1201), 2, 0)}


@pytest.mark.parametrize('name', list(TABLES))
def test_law_and_samples(name):
    rng = np.random.default_rng(11)
    make, law, first = TABLES[name]
    x = make(rng)
    f = np.cos(3. * np.arange(x.size)) * np.sqrt(1. + np.arange(x.size))
    found = find_law(x)
    assert found[:2] == (law, first), found
    if not found[0]:
        return
    q = samples(x, rng)
    out, outside = apply(x, f, found, q)
    assert not outside and np.array_equal(out, np.interp(q, x, f))
    bad = np.concatenate([q[:100], [np.nextafter(x[0], -np.inf), np.nextafter(x[-1], np.inf), np.nan]])
    out, outside = apply(x, f, found, bad)
    assert outside and np.isnan(out[-3:]).all() and np.array_equal(out[:-3], np.interp(q[:100], x, f))


def test_wrong_laws_still_interpolate():
    """A law the table does not follow (here: handed over by the test; in the library: one that passed the check on the knots but guesses worse between
    them) costs steps of the walk, never the result."""
    rng = np.random.default_rng(12)
    x = np.sort(rng.uniform(1., 10., 300))
    x[40:44] = x[40]      # repeated knots as well
    f = rng.standard_normal(x.size)
    q = samples(x, rng, 5000)
    ref = np.interp(q, x, f)
    for law in [(1, 0, x[0], (x.size - 1) / (x[-1] - x[0])), (2, 0, np.log2(x[0]), (x.size - 1) / np.log2(x[-1] / x[0])), (1, 0, 0., 0.), (1, 0, -1e300, 1e300), (2, 0, 50., -3.),
                (1, 5, x[5], 1.), (2, 7, 0., 1e9)]:
        out, outside = apply(x, f, law, q)
        assert not outside and np.array_equal(out, ref), law
    # infinite values in the table: numpy's retry from the right knot and its common-value rule
    f2 = f.copy()
    f2[100], f2[200:202] = np.inf, -np.inf
    with np.errstate(invalid='ignore'):
        ref = np.interp(q, x, f2)
    out, _ = apply(x, f2, find_law(np.linspace(1., 10., 300))[:2] + (x[0], (x.size - 1) / (x[-1] - x[0])), q)
    assert np.array_equal(out, ref, equal_nan=True)
