"""Pin the oracle restatement of the remaining P(k) BAO filters (hinton2017, savgol, ehsavgol, ehpoly,
peakaverage; oracle/bao.py) against golden vectors from the reference (tests/golden/bao2.npz)."""
import numpy as np
import pytest

from oracle import bao as obao
from oracle import interp as oi
from oracle import power as op

RTOL = 1e-9


def pk_fid_callables():
    import oracle.background as ob
    from test_oracle_power_sigma import eh_default_callable
    pk = eh_default_callable(0.)
    p = ob.derived()
    from oracle import sigma as osg

    def pknow(k):     # EH no-wiggle of the default cosmology, sigma8-normalised as Fourier(cosmo, engine='eisenstein_hu_nowiggle') does
        D0 = op.growth_factor(0., p, znorm=0.)
        raw = lambda kk: op.pk_z0(kk, 'eisenstein_hu_nowiggle', sigma8=0.8) * D0**2
        rs = 0.8 / np.sqrt(osg.sigma_r2(8., raw))
        return raw(k) * rs**2
    return pk, pknow


@pytest.mark.parametrize('ic', range(4))
def test_filters_1d(golden, ic):
    g = golden('bao2')
    k, pk, pknow_eh, ratio = g['k'], g['c%d_pk' % ic], g['c%d_pknow_eh' % ic], float(g['c%d_rs_ratio' % ic])
    np.testing.assert_allclose(obao.hinton2017(k, pk)[:, 0], g['c%d_hinton2017_pknow' % ic], rtol=1e-7)
    np.testing.assert_allclose(obao.savgol(k, pk)[:, 0], g['c%d_savgol_pknow' % ic], rtol=RTOL)
    np.testing.assert_allclose(obao.ehsavgol(k, pk, pknow_eh)[:, 0], g['c%d_ehsavgol_pknow' % ic], rtol=RTOL)
    np.testing.assert_allclose(obao.ehpoly(k, pk, pknow_eh, rs_ratio=ratio)[:, 0], g['c%d_ehpoly_pknow' % ic], rtol=RTOL)
    prep = ([g['peakaverage_k_peaks0'], g['peakaverage_k_peaks1']], [tuple(g['peakaverage_pad_peaks0']), tuple(g['peakaverage_pad_peaks1'])])
    np.testing.assert_allclose(obao.peakaverage(k, pk, pknow_eh, prep, rs_ratio=ratio)[:, 0], g['c%d_peakaverage_pknow' % ic], rtol=RTOL)


def test_peakaverage_prepare(golden):
    g = golden('bao2')
    pk_fid, pknow_fid = pk_fid_callables()
    k_peaks, pad_peaks = obao.peakaverage_prepare(g['k'], pk_fid, pknow_fid)
    # The reference's list of maxima ends with a rounding-noise peak at k = 0.967, where the wiggles have died out and the
    # ratio is flat to 1 ulp: not reproducible by construction.  The physical extrema (k < 0.9) and the padding must agree.
    for j in range(2):
        mine, theirs = k_peaks[j], g['peakaverage_k_peaks%d' % j]
        np.testing.assert_allclose(mine[(mine > 1e-2) & (mine < 0.9)], theirs[(theirs > 1e-2) & (theirs < 0.9)], rtol=1e-14)
        assert pad_peaks[j][0] == g['peakaverage_pad_peaks%d' % j][0]
    assert np.isclose(g['peakaverage_k_peaks0'][g['peakaverage_pad_peaks0'][0] + g['peakaverage_pad_peaks0'][1] - 1], 0.96680134)


def test_filters_table(golden):
    """4-column tabulated input: every column filtered; hinton2017 centres its weights on column 0."""
    g = golden('bao2')
    k = g['k']
    tab = oi.pk_interp_2d(g['tab_k'], g['tab_z'], g['tab_pk'])
    pk = tab(k, g['tab_z'])
    np.testing.assert_allclose(obao.hinton2017(k, pk), g['tab_hinton2017_pknow'], rtol=1e-7)
    np.testing.assert_allclose(obao.savgol(k, pk), g['tab_savgol_pknow'], rtol=RTOL)
    np.testing.assert_allclose(obao.ehsavgol(k, pk, g['c2_pknow_eh']), g['tab_ehsavgol_pknow'], rtol=RTOL)
    np.testing.assert_allclose(obao.ehpoly(k, pk, g['c2_pknow_eh'], rs_ratio=float(g['c2_rs_ratio'])), g['tab_ehpoly_pknow'], rtol=RTOL)


BSPLINE_CASES = {'none': (), 'sigma8_np1': ('sigma8',), 'sigma8_sigmad_np1': ('sigma8', 'sigmad')}


@pytest.mark.parametrize('ic', range(4))
def test_bspline_1d(golden, ic):
    """oracle/bao.py: bspline against the reference's outputs (tests/golden/bspline.npz; the `_np1` cases are the reference run with the
    numpy < 2 meaning of its last ``linalg.solve``, see oracle/gen_golden.py: gen_bspline)."""
    g = golden('bspline')
    for name, constraint in BSPLINE_CASES.items():
        got = obao.bspline(g['k'], g['c%d_pk' % ic], g['c%d_pknow_eh' % ic], constraint=constraint)[:, 0]
        np.testing.assert_allclose(got, g['c%d_%s' % (ic, name)], rtol=1e-12, err_msg=name)


def test_bspline_table(golden):
    g = golden('bspline')
    pk = oi.pk_interp_2d(g['tab_k'], g['tab_z'], g['tab_pk'])(g['k'], g['tab_z'])
    for name, constraint in BSPLINE_CASES.items():
        ref = g['tab_' + (name if name.endswith('_np1') else name + '_np1')]
        np.testing.assert_allclose(obao.bspline(g['k'], pk, g['c2_pknow_eh'], constraint=constraint), ref, rtol=1e-11, err_msg=name)
