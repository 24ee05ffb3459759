"""The short forms of exp / log / 10^x / sin / reciprocal / reciprocal root the ALU-bound kernels use (csrc/cp_math.h), through cp_math_eval, against 80-bit
arithmetic: the accuracies their comments state.  (numpy's longdouble functions: 64-bit mantissa, relative error 1e-19.)"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _eval(name, x):
    import torch
    from cosmoprimo_amd import _lib, _device as dv
    dev = torch.device('cuda', 0)
    tx = torch.as_tensor(np.ascontiguousarray(x, dtype='f8'), device=dev)
    ty = torch.empty_like(tx)
    _lib.check(_lib.load().cp_math_eval(_lib.MATH_FUNCTIONS[name], tx.data_ptr(), ty.data_ptr(), tx.numel(), 0, dv.stream_of(dev)))
    return ty.cpu().numpy()


def _rel(got, ref):
    ref = np.asarray(ref, dtype=np.longdouble)
    return float(np.max(np.abs((got.astype(np.longdouble) - ref) / ref)))


def test_exponentials():
    rng = np.random.default_rng(1)
    x = np.concatenate([rng.uniform(-690., 700., 200000), rng.uniform(-5., 5., 200000), [0., -0., 1e-300, -1e-300, 709.7]])      # (results in the normal range)
    ref = np.exp(x.astype(np.longdouble))
    for name, bound in (('exp_mid', 2.3e-16), ('exp_tab', 2.3e-16)):
        assert _rel(_eval(name, x), ref) < bound, name
    # beyond the doubles, and NaN
    assert np.array_equal(_eval('exp_tab', [-800., -1e4, -np.inf]), [0., 0., 0.]) and _eval('exp_tab', [710.])[0] == np.inf and np.isnan(_eval('exp_tab', [np.nan])[0])
    x10 = np.concatenate([rng.uniform(-299., 299., 200000), rng.uniform(-8., 8., 200000)])
    ref10 = np.exp(x10.astype(np.longdouble) * np.log(np.longdouble(10.)))
    for name, bound in (('exp10_mid', 2.3e-16), ('exp10_tab', 2.5e-16)):
        assert _rel(_eval(name, x10), ref10) < bound, name
    assert np.array_equal(_eval('exp10_mid', [-400., 400., -np.inf, np.inf]), [0., np.inf, 0., np.inf])


def test_logarithms():
    rng = np.random.default_rng(2)
    x = np.concatenate([np.exp(rng.uniform(-700., 700., 200000)), rng.uniform(0.5, 2., 200000), rng.uniform(2.7, 30., 100000), [1., 2., 0.5, np.e]])
    ref = np.log(x.astype(np.longdouble))
    got = _eval('log_pos', x)
    nz = ref != 0
    assert _rel(got[nz], ref[nz]) < 2.3e-16 and got[~nz].max() == 0.
    # the table-driven form: absolute error 2e-16 max(1, |log x|) -- the relative error of the above for every argument away from 1
    got = _eval('log_tab', x)
    assert float(np.max(np.abs(got.astype(np.longdouble) - ref) / np.maximum(1., np.abs(ref)))) < 2.3e-16
    away = np.abs(ref) >= 1.
    assert _rel(got[away], ref[away]) < 2.3e-16
    # outside the domain: the library's answers
    out = _eval('log_tab', [0., -1., np.inf, np.nan, 5e-324])
    assert out[0] == -np.inf and np.isnan(out[1]) and out[2] == np.inf and np.isnan(out[3]) and abs(out[4] - np.log(5e-324)) < 1e-12


def test_sine_reciprocal_root():
    rng = np.random.default_rng(3)
    x = np.concatenate([rng.uniform(-1e6, 1e6, 200000), rng.uniform(-10., 10., 200000)])
    ref = np.sin(x.astype(np.longdouble))
    got = _eval('sin_bounded', x)
    assert float(np.max(np.abs(got.astype(np.longdouble) - ref))) < 2.5e-16
    y = np.exp(rng.uniform(-600., 600., 300000)) * rng.choice([-1., 1.], 300000)
    assert _rel(_eval('recip', y), 1. / y.astype(np.longdouble)) < 2.5e-16
    z = np.abs(y)
    assert _rel(_eval('rsqrt_pos', z), 1. / np.sqrt(z.astype(np.longdouble))) < 2.5e-16
