"""CPU: the C-ABI library loads and exports every symbol include/cosmoprimo_amd.h declares (no compute calls)."""
import os
import re

import numpy as np
import pytest

from conftest import ROOT
from cosmoprimo_amd import _lib


def header_symbols():
    text = open(os.path.join(ROOT, 'include', 'cosmoprimo_amd.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(cp_[a-z_0-9]+)\s*\(', text)))


def test_library_exports_header():
    lib = _lib.load()
    names = header_symbols()
    assert len(names) >= 10
    for name in names:
        assert hasattr(lib, name), name
        assert name in _lib.SIGNATURES, 'binding missing for {}'.format(name)
    assert lib.cp_abi_version() == _lib.ABI_VERSION


def test_error_mapping():
    lib = _lib.load()
    import ctypes
    handle = ctypes.c_void_p()
    pre = np.ones(8)
    with pytest.raises(ValueError):   # npad not a power of two -> CP_EINVAL -> ValueError (reference fftlog.py:155-159)
        _lib.check(lib.cp_fftlog_plan_create(ctypes.byref(handle), 3, 6, 1, _lib.as_double_p(pre), _lib.as_double_p(pre), _lib.as_double_p(pre), 0))
    assert b'power of two' in lib.cp_last_error()
    with pytest.raises(NotImplementedError):  # beyond the supported range (2^24) -> CP_EUNSUPPORTED
        _lib.check(lib.cp_fftlog_plan_create(ctypes.byref(handle), 10000, 1 << 25, 1, _lib.as_double_p(pre), _lib.as_double_p(pre), _lib.as_double_p(pre), 0))
    with pytest.raises(ValueError):
        _lib.kernel_eval(99, 0., np.ones(2, dtype='c16'))
    # argument checks of the entry points added later in the round come before any device call
    for call in (lambda: lib.cp_interp_linear(None, None, 0, None, None, 4, 0, None), lambda: lib.cp_interp_linear(None, None, 8, None, None, 4, 0, None),
                 lambda: lib.cp_rows_screen(None, -1, 8, 0, None, None, 0, None), lambda: lib.cp_rows_screen(None, 4, 8, 0, None, None, 0, None),
                 lambda: lib.cp_wallish_box(None, 4, 64, 40, 5, -10, 20, None, 0, None), lambda: lib.cp_wallish_box(None, 4, 2048, 20, 5, -10, 20, None, 0, None),
                 lambda: lib.cp_dst_execute(None, None, None, 4, 0, 0, None)):
        with pytest.raises(ValueError):
            _lib.check(call())
    assert lib.cp_interp_linear(None, None, 8, None, None, 0, 0, None) == 0 and lib.cp_rows_screen(None, 0, 8, 0, None, None, 0, None) == 0     # nothing to do


def test_error_mapping_of_the_table_and_distance_entry_points():
    """Argument checks of cp_interp_table_* / cp_distance_from_radial / cp_spline_plan_create come before any device call."""
    import ctypes
    lib = _lib.load()
    handle = ctypes.c_void_p()
    x, f = np.array([0., 2., 1.]), np.zeros(3)
    with pytest.raises(ValueError):      # x not ascending
        _lib.check(lib.cp_interp_table_create(ctypes.byref(handle), 3, _lib.as_double_p(x), _lib.as_double_p(f), 0))
    assert b'ascending' in lib.cp_last_error() and not handle.value
    with pytest.raises(ValueError):      # no rows
        _lib.check(lib.cp_interp_table_create(ctypes.byref(handle), 0, _lib.as_double_p(x), _lib.as_double_p(f), 0))
    with pytest.raises(ValueError):      # null table
        _lib.check(lib.cp_interp_table_apply(None, None, None, 4, None, None))
    with pytest.raises(ValueError):
        _lib.check(lib.cp_interp_table_apply_f32(None, None, None, 4, None, None))
    with pytest.raises(ValueError):
        _lib.check(lib.cp_interp_table_law(None, None, None))
    assert lib.cp_interp_table_destroy(None) == 0
    with pytest.raises(ValueError):      # comoving_radial_distance (kind 0) is not a derived distance
        _lib.check(lib.cp_distance_from_radial(None, None, 4, 0., 0, None, 0, None))
    with pytest.raises(ValueError):      # negative size
        _lib.check(lib.cp_distance_from_radial(None, None, -1, 0., 3, None, 0, None))
    assert lib.cp_distance_from_radial(None, None, 0, 0., 3, None, 0, None) == 0      # nothing to do
    knots, queries = np.linspace(0., 1., 600), np.linspace(0., 1., 1 << 20)
    with pytest.raises(NotImplementedError):      # 6.3e8 weights: the dense staging of the operator would take 5 GB on the host
        _lib.check(lib.cp_spline_plan_create(ctypes.byref(handle), knots.size, _lib.as_double_p(knots), queries.size, _lib.as_double_p(queries), _lib.SPLINE_BC['natural'], 0, 0, 0))
    assert b'2^29' in lib.cp_last_error()
    # ... and every other *_create that stages caller-sized arrays on the host refuses sizes beyond its limit before it allocates or touches a device
    one = np.ones(8)
    with pytest.raises(NotImplementedError):      # 2^20 x 2^10 weights of a dense operator (the array itself is never read)
        _lib.check(lib.cp_linop_plan_create(ctypes.byref(handle), 1 << 20, 1 << 10, _lib.as_double_p(one), 0))
    with pytest.raises(NotImplementedError):      # 2^27 queries of a row spline
        _lib.check(lib.cp_spline_rows_plan_create(ctypes.byref(handle), 8, _lib.as_double_p(np.arange(8.)), _lib.SPLINE_BC['natural'], 0, 1 << 27, _lib.as_double_p(one), 0))
    with pytest.raises(NotImplementedError):      # a table of 2^28 rows
        _lib.check(lib.cp_interp_table_create(ctypes.byref(handle), 1 << 28, _lib.as_double_p(one), _lib.as_double_p(one), 0))
    with pytest.raises(NotImplementedError):      # 2^13 transforms in parallel of padded size 2^16: 2^29 table entries
        _lib.check(lib.cp_fftlog_plan_create(ctypes.byref(handle), 1 << 15, 1 << 16, 1 << 13, _lib.as_double_p(one), _lib.as_double_p(one), _lib.as_double_p(one), 0))
    assert not handle.value


def test_error_mapping_of_the_prefiltered_spline_plan():
    """Everything cp_geospline_plan_create_prefiltered refuses, it refuses before its first device call (this box has no device): sizes other than
    1024 -> 2048, knots that are not geometric, a postfactor that is no power law, radii near the ends of the knots or too far apart; and
    cp_fftlog_geospline_execute wants exactly one transform."""
    import ctypes
    lib = _lib.load()
    handle = ctypes.c_void_p()
    n, npad = 1024, 2048
    knots = np.geomspace(1e-2, 1e7, n)
    j = np.arange(npad)
    pre, post, u = np.ones(npad), 3. * 0.97**j, np.ones(2 * (npad // 2 + 1))
    radii = np.geomspace(1., 100., 16)

    def create(n=n, npad=npad, pre=pre, post=post, u=u, knots=knots, radii=radii):
        return lib.cp_geospline_plan_create_prefiltered(ctypes.byref(handle), n, npad, _lib.as_double_p(pre), _lib.as_double_p(post), _lib.as_double_p(u),
                                                        _lib.as_double_p(knots), _lib.as_double_p(radii), radii.size, 0)

    assert create(n=512, npad=1024) == _lib.CP_EUNSUPPORTED and b'1024 samples' in lib.cp_last_error()
    assert create(knots=np.linspace(1., 2., n)) == _lib.CP_EUNSUPPORTED and b'geometric' in lib.cp_last_error()
    assert create(post=post * (1. + 1e-6 * np.sin(j))) == _lib.CP_EUNSUPPORTED and b'power law' in lib.cp_last_error()
    assert create(post=np.zeros(npad)) == _lib.CP_EUNSUPPORTED
    assert create(radii=np.array([knots[3] * 1.01, 8.])) == _lib.CP_EUNSUPPORTED and b'within 32 knots' in lib.cp_last_error()
    assert create(radii=np.array([8., knots[-5]])) == _lib.CP_EUNSUPPORTED
    assert create(radii=np.geomspace(knots[40], knots[900], 8)) == _lib.CP_EUNSUPPORTED and b'span' in lib.cp_last_error()
    assert create(radii=np.array([1e-9, 1e12])) == _lib.CP_EUNSUPPORTED and b'no query inside' in lib.cp_last_error()
    assert create(radii=np.geomspace(1., 100., 513)) == _lib.CP_EUNSUPPORTED
    assert lib.cp_geospline_plan_create_prefiltered(ctypes.byref(handle), n, npad, None, None, None, None, None, 4, 0) == _lib.CP_EINVAL
    assert not handle.value
    with pytest.raises(ValueError):      # no spline plan at all
        _lib.check(lib.cp_fftlog_geospline_execute(None, None, ctypes.c_void_p(8), ctypes.c_void_p(8), 4, 0, 0, None))
    assert lib.cp_fftlog_geospline_execute(None, None, None, None, 0, 0, 0, None) == 0      # nothing to do


def test_loggamma_vs_scipy_golden(golden):
    # G2: scipy.special.loggamma / gamma on the kernels' arguments + stress grid
    g = golden('loggamma')
    out = _lib.loggamma(g['z'])
    ref = g['loggamma']
    assert np.all(np.abs(out - ref) <= 4e-16 * np.maximum(np.abs(ref), 1.))
    outg = _lib.gamma(g['zg'])
    assert np.all(np.abs(outg - g['gamma']) <= 2e-15 * np.abs(g['gamma']))


def test_loggamma_special_points():
    from scipy.special import loggamma
    z = np.array([1., 2., 0.5, 1.05 + 0.02j, 2.1 - 0.05j, -0.5 + 1e-3j, -3.3 - 2j, 6.9 + 6.9j, 7.1 + 0.1j, 0.05 + 0.01j, -2.5 + 0j, 30. + 0j,
                  1e-8 + 0j, -1e-8 + 1e-9j, 3 + 1e-300j])
    out = _lib.loggamma(z)
    ref = loggamma(z)
    assert np.all(np.abs(out - ref) <= 1e-15 * np.maximum(np.abs(ref), 1.)), np.abs(out - ref)
    assert np.all(np.isnan(_lib.loggamma(np.array([0., -1., -2.])).real))  # poles


@pytest.mark.gpu
def test_c_consumer(tmp_path):
    """A consumer of the C ABI without Python or torch in the data path (tests/abi/consumer.cpp): HIP-runtime buffers, plain pointers; checked
    against a direct O(N^2) evaluation of the reference arithmetic inside the program."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / 'consumer')
    libdir = os.path.join(root, 'cosmoprimo_amd')
    subprocess.run(['hipcc', '--offload-arch=gfx950', '-O2', '-I', os.path.join(root, 'include'), '-o', exe, os.path.join(root, 'tests', 'abi', 'consumer.cpp'),
                    '-L', libdir, '-lcosmoprimo_amd', '-Wl,-rpath,' + libdir], check=True, capture_output=True, timeout=300)
    res = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert res.returncode == 0, res.stdout + res.stderr
    assert 'OK (ABI' in res.stdout
