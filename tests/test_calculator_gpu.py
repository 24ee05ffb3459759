"""Batch driver ``emulators.get_calculator`` (reference cosmoprimo/emulators/__init__.py:11-60) against the reference's
own calculator output (tests/golden/calculator.npz, made by oracle/gen_golden.py calculator) and batched-vs-one-at-a-time consistency."""
import warnings

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _cases():
    from oracle.gen_golden import CALCULATOR_CASES, CALCULATOR_PK_STRIDE
    return CALCULATOR_CASES, CALCULATOR_PK_STRIDE


def test_calculator_matches_reference(golden):
    import cosmoprimo_amd as cp
    from cosmoprimo_amd.emulators import get_calculator
    warnings.simplefilter('ignore')
    g = golden('calculator')
    cases, (sk, sz) = _cases()
    for i, (engine, base, params) in enumerate(cases):
        out = get_calculator(cp.Cosmology(engine=engine, **base))(**params)
        expected = {name[len('c%d:' % i):]: g[name] for name in g if name.startswith('c%d:' % i)}
        assert set(expected) <= set(out), set(expected) - set(out)
        for name, ref in expected.items():
            val = np.asarray(out[name])
            if name.startswith('fourier.pk.'):
                val = val[::sk, ::sz]
            assert val.shape == ref.shape, (engine, name, val.shape, ref.shape)
            # E(z), distances, densities, P(k): pointwise 1e-10; time: the reference integrates 1/(a E) with its own
            # quadrature to ~1e-7 (see test_background_gpu), the massive-neutrino momenta are splines of its 10-point Gauss-Laguerre rule
            rtol = {'background.time': 2e-6, 'background.rho_ncdm': 1e-8, 'background.p_ncdm': 1e-8}.get(name, 1e-10)
            if 'variants' in engine and name.startswith('fourier.pk'):
                rtol = 1e-8      # sigma8 normalisation of the reference goes through its FFTLog + spline on P(k) tables
            np.testing.assert_allclose(val, ref, rtol=rtol, atol=0., err_msg='%s %s' % (engine, name))


def test_calculator_batch():
    import cosmoprimo_amd as cp
    from cosmoprimo_amd.emulators import get_calculator, CalculatorComputationError
    warnings.simplefilter('ignore')
    rng = np.random.default_rng(7)
    nb = 96
    params = dict(Omega_m=rng.uniform(0.25, 0.4, nb), h=rng.uniform(0.6, 0.8, nb), n_s=rng.uniform(0.92, 1., nb), w0_fld=rng.uniform(-1.2, -0.8, nb))
    calc = get_calculator(cp.Cosmology(engine='eisenstein_hu'))
    out = calc(**params)
    assert out['background.comoving_radial_distance'].shape == (nb, 256) and out['fourier.pk.delta_m.delta_m'].shape == (nb, 422, 30)
    assert out['thermodynamics.rs_drag'].shape == (nb,) and out['primordial.A_s'].shape == (nb,)
    for row in (0, 41, nb - 1):
        one = calc(**{name: float(value[row]) for name, value in params.items()})
        assert set(one) == set(out)
        for name, value in one.items():
            batched = out[name] if name in ('fourier.k', 'fourier.z', 'background.z') else out[name][row]
            np.testing.assert_allclose(batched, value, rtol=1e-12, atol=0., err_msg=name)
    # a batch of cosmologies with one massive species: (B, N_ncdm, nz) densities, rows equal to one-at-a-time calls
    calc_nu = get_calculator(cp.Cosmology(engine='eisenstein_hu_nowiggle_variants', m_ncdm=[0.06]), section=['background', 'thermodynamics'])
    Om = np.linspace(0.27, 0.33, 5)
    out = calc_nu(Omega_m=Om)
    assert out['background.rho_ncdm'].shape == (5, 1, 256)
    one = calc_nu(Omega_m=float(Om[3]))
    for name in ('background.rho_ncdm', 'background.p_ncdm', 'background.comoving_radial_distance', 'background.time', 'thermodynamics.rs_drag'):
        np.testing.assert_allclose(out[name][3], one[name], rtol=1e-12, atol=0., err_msg=name)
    # section selection, pass-through of anything that is not a Cosmology, error translation (reference :26, 52-53)
    only = get_calculator(cp.Cosmology(engine='eisenstein_hu'), section='background')(Omega_m=0.3)
    assert all(name.startswith('background.') for name in only) and len(only) == 6
    assert get_calculator(calc) is calc
    with pytest.raises(CalculatorComputationError):
        calc(w0_fld=0.5, wa_fld=0.)
