"""GPU parity of the remaining P(k) BAO filters (hinton2017, savgol, ehsavgol, ehpoly, peakaverage) against
golden vectors from the reference (tests/golden/bao2.npz) and the oracle restatement (oracle/bao.py).  Tolerance 1e-8 on pknow
(1e-7 for hinton2017, whose degree-12 normal equations the reference inverts explicitly)."""
import warnings

import numpy as np
import pytest

from oracle import bao as obao
from oracle.gen_golden import BAO_PARAMS

pytestmark = pytest.mark.gpu
ENGINES = ['hinton2017', 'savgol', 'ehsavgol', 'ehpoly', 'peakaverage']
RTOL = {'hinton2017': 1e-7}


@pytest.fixture(scope='module')
def cp():
    import torch
    assert torch.cuda.is_available(), 'GPU tests need a ROCm device'
    import cosmoprimo_amd
    return cosmoprimo_amd


@pytest.mark.parametrize('ic', range(4))
def test_filters_1d(cp, golden, ic):
    g = golden('bao2')
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        fid = cp.Cosmology(engine='eisenstein_hu')
        cosmo = cp.Cosmology(engine='eisenstein_hu', **BAO_PARAMS[ic])
        interp = cosmo.get_fourier().pk_interpolator().to_1d(z=0.)
        for eng in ENGINES:
            f = cp.PowerSpectrumBAOFilter(interp, engine=eng, cosmo=cosmo, cosmo_fid=fid)
            np.testing.assert_allclose(f.k, g['k'], rtol=1e-14)
            ref = g['c%d_%s_pknow' % (ic, eng)]
            if eng == 'peakaverage':      # the package's own knot search (tie at the end of the series settled by rule): the reference's lists
                for j in range(2):
                    np.testing.assert_allclose(f.k_peaks[j], g['peakaverage_k_peaks%d' % j], rtol=1e-14)
                    assert tuple(f.pad_peaks[j]) == tuple(g['peakaverage_pad_peaks%d' % j])
            np.testing.assert_allclose(f.pknow, ref, rtol=RTOL.get(eng, 1e-8), err_msg=eng)
            assert f.pknow.shape == f.pk.shape == (1024,) and np.abs(f.wiggles - 1.).max() < 0.2
    np.testing.assert_allclose(f.rs_drag_ratio(), g['c%d_rs_ratio' % ic], rtol=1e-10)


def test_filters_table(cp, golden):
    g = golden('bao2')
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        fid = cp.Cosmology(engine='eisenstein_hu')
        cosmo = cp.Cosmology(engine='eisenstein_hu', **BAO_PARAMS[2])
        tab = cp.PowerSpectrumInterpolator2D(g['tab_k'], g['tab_z'], g['tab_pk'])
        for eng in ENGINES:
            kw = dict(cosmo=cosmo, cosmo_fid=fid) if eng in ('ehsavgol', 'ehpoly', 'peakaverage') else {}
            f = cp.PowerSpectrumBAOFilter(tab, engine=eng, **kw)
            assert f.pknow.shape == (1024, 4)
            np.testing.assert_allclose(f.pknow, g['tab_%s_pknow' % eng], rtol=RTOL.get(eng, 1e-8), err_msg=eng)
            if eng in ('savgol', 'hinton2017'):     # same input: device operator == oracle arithmetic
                np.testing.assert_allclose(f.pknow, getattr(obao, eng)(f.k, f.pk), rtol=RTOL.get(eng, 1e-8))
    with pytest.raises(ValueError):
        cp.PowerSpectrumBAOFilter(tab, engine='peakaverage', cosmo=cosmo)      # cosmo_fid is mandatory
    with pytest.raises(ValueError):
        cp.PowerSpectrumBAOFilter(tab, engine='no_such_filter')


BSPLINE_CASES = {'none': (), 'sigma8_np1': ('sigma8',), 'sigma8_sigmad_np1': ('sigma8', 'sigmad')}


@pytest.mark.parametrize('ic', range(4))
def test_bspline_1d(cp, golden, ic):
    """`bspline` (reference bao_filter.py:583-688) against the reference's own outputs: as it runs here without constraint, and with the
    numpy < 2 meaning of its final ``linalg.solve`` for the constrained cases (oracle/gen_golden.py: gen_bspline)."""
    g = golden('bspline')
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        cosmo = cp.Cosmology(engine='eisenstein_hu', **BAO_PARAMS[ic])
        interp = cosmo.get_fourier().pk_interpolator().to_1d(z=0.)
        for name, constraint in BSPLINE_CASES.items():
            f = cp.PowerSpectrumBAOFilter(interp, engine='bspline', cosmo=cosmo, constraint=constraint)
            np.testing.assert_allclose(f.k, g['k'], rtol=1e-14)
            np.testing.assert_allclose(f.pknow, g['c%d_%s' % (ic, name)], rtol=1e-8, err_msg=name)
            assert f.pknow.shape == (1024,)
        f = cp.PowerSpectrumBAOFilter(interp, engine='bspline', cosmo=cosmo)       # default: sigma8 is kept
        np.testing.assert_allclose(f.pknow, g['c%d_sigma8_np1' % ic], rtol=1e-8)
    with pytest.raises(ValueError):
        cp.PowerSpectrumBAOFilter(interp, engine='bspline', cosmo=cosmo, constraint=('sigma12',))


def test_bspline_table(cp, golden):
    g = golden('bspline')
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        cosmo = cp.Cosmology(engine='eisenstein_hu', **BAO_PARAMS[2])
        tab = cp.PowerSpectrumInterpolator2D(g['tab_k'], g['tab_z'], g['tab_pk'])
        for name, constraint in BSPLINE_CASES.items():
            f = cp.PowerSpectrumBAOFilter(tab, engine='bspline', cosmo=cosmo, constraint=constraint)
            ref = g['tab_' + (name if name.endswith('_np1') else name + '_np1')]
            assert f.pknow.shape == ref.shape == (1024, 4)
            np.testing.assert_allclose(f.pknow, ref, rtol=1e-8, err_msg=name)
            # same input through the oracle's per-column arithmetic
            pknow_eh = np.asarray(cp.Fourier(cosmo, engine='eisenstein_hu_nowiggle', set_engine=False).pk_interpolator()(f.k, z=0.))
            np.testing.assert_allclose(f.pknow, obao.bspline(f.k, f.pk, pknow_eh, constraint=constraint), rtol=1e-8, err_msg=name)
