"""GPU parity of the wallish2018 and brieden2022 BAO filters against golden vectors produced by the reference (G6).
Tolerance: relative <= 1e-9 on pknow."""
import warnings

import numpy as np
import pytest

from oracle.gen_golden import BAO_PARAMS

pytestmark = pytest.mark.gpu
RTOL = 1e-9


@pytest.fixture(scope='module')
def cp():
    import torch
    assert torch.cuda.is_available()
    import cosmoprimo_amd
    warnings.simplefilter('ignore')
    return cosmoprimo_amd


def test_wallish2018(cp, golden):
    from cosmoprimo_amd.bao_filter import PowerSpectrumBAOFilter
    g = golden('bao')
    fid = cp.Cosmology(engine='eisenstein_hu')
    for i, par in enumerate(BAO_PARAMS):
        cosmo = cp.Cosmology(engine='eisenstein_hu', **par)
        interp = cosmo.get_fourier().pk_interpolator().to_1d(z=0.)
        f = PowerSpectrumBAOFilter(interp, engine='wallish2018', cosmo=cosmo, cosmo_fid=fid)
        assert f.pknow.shape == (1024,) and f.k.shape == (1024,)
        np.testing.assert_allclose(f.k, g['k'], rtol=1e-14)
        np.testing.assert_allclose(f.pk, g['c%d_pk' % i], rtol=1e-10)
        np.testing.assert_allclose(f.pknow, g['c%d_wallish_pknow' % i], rtol=RTOL)
        if i == 0:   # intermediates the reference keeps (bao_filter.py:383-386, 407-408)
            # entry by entry at 1e-9, above a noise floor of 1e-14 of the scale of the sequence the entry is computed FROM (an entry carries the rounding of
            # the whole transform: measured 2e-16 of that scale; the entries fall over ten decades along a sequence, the second derivatives over fourteen:
            # their smallest, 3e-10 of the largest, are still held to 1e-5 of themselves)
            scale = np.abs(g['c0_wallish_even'][:, 0]).max()
            for got, ref in ((f._even_now.cpu().numpy()[0], g['c0_wallish_even_now'][:, 0]), (f._odd_now.cpu().numpy()[0], g['c0_wallish_odd_now'][:, 0]),
                             (f._dd[0].cpu().numpy()[0], g['c0_wallish_dd_even'][:, 0])):
                np.testing.assert_allclose(got, ref, rtol=1e-9, atol=1e-14 * scale)
        if i == 1:
            f2 = PowerSpectrumBAOFilter(interp, engine='wallish2018')
            np.testing.assert_allclose(f2.pknow, g['c1_wallish_pknow_nocosmo'], rtol=RTOL)
            assert np.allclose(f.wiggles, f.pk / f.pknow)


def test_brieden2022(cp, golden):
    from cosmoprimo_amd.bao_filter import PowerSpectrumBAOFilter
    g = golden('bao')
    fid = cp.Cosmology(engine='eisenstein_hu')
    for i, par in enumerate(BAO_PARAMS):
        cosmo = cp.Cosmology(engine='eisenstein_hu', **par)
        interp = cosmo.get_fourier().pk_interpolator().to_1d(z=0.)
        f = PowerSpectrumBAOFilter(interp, engine="brieden2022", cosmo=cosmo, cosmo_fid=fid)
        np.testing.assert_allclose(f.rs_drag_ratio(), g['c%d_rs_drag_ratio' % i], rtol=1e-12)
        if i == 0:
            np.testing.assert_allclose(f.k_fid, g['brieden_k_fid'], rtol=1e-14)
            np.testing.assert_allclose(f.pknow_correction, g['brieden_pknow_correction'], rtol=1e-10)
            np.testing.assert_allclose(f.ratio_fid, g['brieden_ratio_fid'], rtol=1e-10)

        # envelope knots from the package's own search, the end-of-series tie settled by rule (bao_filter._wiggle_extrema): the
        # reference's lists for this fiducial cosmology, knot 339 included -- nothing is injected
        assert np.array_equal(f.ik_fid_peaks[0], g['brieden_peaks_high']) and np.array_equal(f.ik_fid_peaks[1], g['brieden_peaks_low'])
        if i == 0:
            np.testing.assert_allclose(f.ratio_now_fid, g['brieden_ratio_now_fid'], rtol=1e-10)
        f(interp, cosmo=cosmo)
        np.testing.assert_allclose(f.pknow, g['c%d_brieden_pknow' % i], rtol=RTOL)
        if i == 1:
            f2 = PowerSpectrumBAOFilter(interp, engine='brieden2022', cosmo_fid=fid)
            np.testing.assert_allclose(f2.pknow, g['c1_brieden_pknow_nocosmo'], rtol=RTOL)
    with pytest.raises(ValueError):
        PowerSpectrumBAOFilter(interp, engine='brieden2022')
    with pytest.raises(ValueError):
        PowerSpectrumBAOFilter(interp, engine='nope')


def test_tabulated_2d_input(cp, golden):
    """2D (k, z) table with 4 columns: the filter runs on all columns at once; 2D result == per-column 1D result (reference test_2d_pk)."""
    from cosmoprimo_amd.bao_filter import PowerSpectrumBAOFilter
    g = golden('bao')
    fid = cp.Cosmology(engine='eisenstein_hu')
    cosmo = cp.Cosmology(engine='eisenstein_hu', **BAO_PARAMS[3])
    tab = cp.PowerSpectrumInterpolator2D(g['tab_k'], g['tab_z'], g['tab_pk'])
    w = PowerSpectrumBAOFilter(tab, engine='wallish2018', cosmo=cosmo, cosmo_fid=fid)
    assert w.pknow.shape == (1024, 4)
    np.testing.assert_allclose(w.pknow, g['tab_wallish_pknow'], rtol=RTOL)
    b = PowerSpectrumBAOFilter(tab, engine='brieden2022', cosmo=cosmo, cosmo_fid=fid)      # envelope knots: the package's own search
    np.testing.assert_allclose(b.pknow, g['tab_brieden_pknow'], rtol=RTOL)
    for iz in range(2):
        one = PowerSpectrumBAOFilter(tab.to_1d(z=g['tab_z'][iz]), engine='wallish2018', cosmo=cosmo, cosmo_fid=fid)
        assert np.allclose(one.pknow, w.pknow[:, iz], rtol=1e-6, atol=1e-6)   # reference tests/test_bao_filter.py:117-136


def test_batched_cosmologies(cp, golden):
    """BASELINE config 4 shape: the 4 golden cosmologies as ONE batched Cosmology; both filters run on all vectors at once
    (brieden2022 with one rs_drag ratio per vector) and must reproduce the per-cosmology reference results."""
    from cosmoprimo_amd.bao_filter import PowerSpectrumBAOFilter
    g = golden('bao')
    fid = cp.Cosmology(engine='eisenstein_hu')
    defaults = dict(Omega_m=0.3, Omega_b=0.05, h=0.7, n_s=0.96, sigma8=0.8)
    par = {name: np.array([p.get(name, defaults[name]) for p in BAO_PARAMS]) for name in defaults}
    cosmo = cp.Cosmology(engine='eisenstein_hu', **par)
    interp = cosmo.get_fourier().pk_interpolator(z=np.array([0.]))
    w = PowerSpectrumBAOFilter(interp, engine='wallish2018', cosmo=cosmo, cosmo_fid=fid)
    assert w.pknow.shape == (4, 1024, 1)
    b = PowerSpectrumBAOFilter(interp, engine='brieden2022', cosmo=cosmo, cosmo_fid=fid)      # envelope knots: the package's own search
    for i in range(4):
        np.testing.assert_allclose(w.pknow[i], g['c%d_wallish_pknow_2d' % i], rtol=RTOL)
        np.testing.assert_allclose(b.pknow[i], g['c%d_brieden_pknow_2d' % i], rtol=RTOL)


def test_brieden_resampling_in_one_kernel(cp):
    """brieden2022 over a batch: cp_brieden_smooth (the same with ratio and envelope formed inside, from P at the extrema of the fiducial wiggles only)
    and cp_brieden_resample (the knots of a cosmology are a geometric grid: second derivatives by two recursions per lane, the
    extrapolated knots of _pad_log as a closed-form correction at either end, 10^x into the k_fid range of P: a wave per cosmology) against the three
    kernels they replace (knot-major arrays, the general per-column spline, the final pass), for batches that do and do not fill the waves of a workgroup,
    rs_drag ratios on both sides of 1 (queries shifted either way, into the long intervals at the ends), a cosmology equal to the fiducial one, and
    fewer wavenumbers (4 knots per lane)."""
    from cosmoprimo_amd import bao_filter as bf
    from cosmoprimo_amd.bao_filter import PowerSpectrumBAOFilter
    fid = cp.Cosmology(engine='eisenstein_hu')
    rng = np.random.default_rng(5)
    for nb, nk in ((1, 1024), (7, 1024), (130, 1024), (9, 640), (5, 1500)):      # (1500: 8 samples per lane, the operator too large for LDS)
        par = dict(Omega_m=rng.uniform(0.24, 0.40, nb), Omega_b=rng.uniform(0.04, 0.06, nb), h=rng.uniform(0.6, 0.8, nb), n_s=rng.uniform(0.92, 1., nb))
        if nb > 1:
            par = {name: np.concatenate([[fid[name]], v[1:]]) for name, v in par.items()}
        cosmo = cp.Cosmology(engine='eisenstein_hu', sigma8=0.8, **par)
        interp = cosmo.get_fourier().pk_interpolator(z=np.array([0.]))
        saved = bf._RESAMPLE_IN_ONE_KERNEL
        try:
            got = []
            for scheme in (2, 1, 0):      # everything behind the two P(k) evaluations as one kernel | ratio, operator, re-sampling kernel | ... , three kernels
                bf._RESAMPLE_IN_ONE_KERNEL = scheme
                got.append(np.asarray(PowerSpectrumBAOFilter(interp, engine='brieden2022', cosmo=cosmo, cosmo_fid=fid, nk=nk).pknow))
        finally:
            bf._RESAMPLE_IN_ONE_KERNEL = saved
        for one in got[:2]:
            assert one.shape == got[2].shape == (nb, nk, 1) and np.isfinite(one).all()
            np.testing.assert_allclose(one, got[2], rtol=1e-11)


def test_brieden_resample_against_the_oracle():
    """cp_brieden_resample on its own, through the C ABI, against the oracle's restatement of what it replaces (oracle/interp.py: pad_log +
    natural spline of log10 P on log10 k, interpolator.py:42-87, 345-350; bao_filter.py:500-509): smooth and wiggly samples, rs_drag ratios on
    both sides of 1 and exactly 1 (every query on a knot), 3 to 8 samples per lane, batches that do not fill a workgroup; a k_fid that is not a
    geometric grid gives NaN over its range, nothing else."""
    import torch
    from oracle import interp as ointerp
    from cosmoprimo_amd import _lib, _device as dv
    lib = _lib.load()
    dev = torch.device('cuda', 0)
    st = dv.stream_of(dev)
    rng = np.random.default_rng(11)
    up = lambda a: torch.as_tensor(np.ascontiguousarray(a, dtype='f8'), device=dev)      # noqa: E731
    for n, nb in ((341, 9), (129, 3), (200, 5), (512, 2), (450, 1)):
        nk, first = n + 300, 137
        k = np.geomspace(1e-5, 50., nk)
        k_fid = k[first:first + n]
        rescale = np.concatenate([[1.], rng.uniform(0.85, 1.2, nb - 1)])
        smooth = 2e4 * (k_fid / 0.02)**0.96 / (1. + (k_fid / 0.02)**2)**1.3
        pknow = smooth[None, :] * rng.uniform(0.5, 2., (nb, 1))
        envelope = 1. + 0.05 * np.sin(k_fid[None, :] * rng.uniform(80., 120., (nb, 1))) * np.exp(-(k_fid / 0.3)**2)
        ratio_now_fid = 1. + 0.01 * np.cos(np.log(k_fid))
        pk = rng.uniform(1., 2., (nb, nk))
        out = torch.full((nb, nk), -1., dtype=torch.float64, device=dev)
        args = [up(envelope), up(pknow), up(ratio_now_fid), up(k_fid), up(np.log10(k_fid)), up(rescale), up(pk)]
        _lib.check(lib.cp_brieden_resample(args[0].data_ptr(), args[1].data_ptr(), args[2].data_ptr(), args[3].data_ptr(), args[4].data_ptr(), args[5].data_ptr(),
                                           1e-7, 1e2, args[6].data_ptr(), out.data_ptr(), nb, n, nk, first, 0, st))
        got = out.cpu().numpy()
        ref = pk.copy()
        for c in range(nb):
            ref[c, first:first + n] = ointerp.pk_interp_1d(k_fid / rescale[c], envelope[c] * pknow[c] * ratio_now_fid, 1e-7, 1e2)(k_fid)
        assert np.isfinite(got).all()
        assert np.array_equal(got[:, :first], pk[:, :first]) and np.array_equal(got[:, first + n:], pk[:, first + n:])
        np.testing.assert_allclose(got, ref, rtol=2e-12)
        np.testing.assert_allclose(got[0, first:first + n], envelope[0] * pknow[0] * ratio_now_fid, rtol=2e-14)      # rescale 1: the samples themselves
    # not a geometric grid: one knot moved by a thousandth of a step
    bent = np.log10(k_fid).copy()
    bent[n // 2] += 1e-3 * (bent[1] - bent[0])
    args[4] = up(bent)
    _lib.check(lib.cp_brieden_resample(args[0].data_ptr(), args[1].data_ptr(), args[2].data_ptr(), args[3].data_ptr(), args[4].data_ptr(), args[5].data_ptr(),
                                       1e-7, 1e2, args[6].data_ptr(), out.data_ptr(), nb, n, nk, first, 0, st))
    got = out.cpu().numpy()
    assert np.isnan(got[:, first:first + n]).all() and np.array_equal(got[:, :first], pk[:, :first]) and np.array_equal(got[:, first + n:], pk[:, first + n:])


def test_brieden_one_kernel_refusals():
    """cp_brieden_smooth / cp_brieden_resample: sizes outside the kernel (fewer than 129 or more than 512 samples per cosmology: CP_EUNSUPPORTED, the caller
    then takes the three kernels), more extrema than lanes, null pointers, a range that does not fit the rows (CP_EINVAL); an empty batch is a success."""
    import torch
    from cosmoprimo_amd import _lib, _device as dv
    lib = _lib.load()
    dev = torch.device('cuda', 0)
    st = dv.stream_of(dev)
    buf = torch.ones(4096, dtype=torch.float64, device=dev)
    idx = torch.zeros(64, dtype=torch.int32, device=dev)
    p = buf.data_ptr()

    def smooth(nb, n, nk, first, npk=23, pk=p):
        return lib.cp_brieden_smooth(p, p, p, p, p, idx.data_ptr(), p, npk, p, p, p, p, 1e-5, 10., pk, p, nb, n, nk, first, 0, st)

    def resample(nb, n, nk, first, pk=p):
        return lib.cp_brieden_resample(p, p, p, p, p, p, 1e-5, 10., pk, p, nb, n, nk, first, 0, st)

    for call in (smooth, resample):
        assert call(0, 341, 1024, 300) == _lib.CP_OK
        assert call(1, 128, 1024, 300) == _lib.CP_EUNSUPPORTED and call(1, 513, 1024, 300) == _lib.CP_EUNSUPPORTED
        assert call(1, 341, 1024, 700) == _lib.CP_EINVAL and call(-1, 341, 1024, 300) == _lib.CP_EINVAL and call(1, 341, 1024, -1) == _lib.CP_EINVAL
        assert call(1, 341, 1024, 300, pk=None) == _lib.CP_EINVAL
    assert smooth(1, 341, 1024, 300, npk=65) == _lib.CP_EINVAL and smooth(1, 341, 1024, 300, npk=0) == _lib.CP_EINVAL
    torch.cuda.synchronize(dev)


def test_wallish_box_kernel():
    """cp_wallish_box against ndarray.argmax on the reference's ranges (bao_filter.py:390-394): first index on ties, NaN as the maximum, the
    index-0 convention for an empty second range."""
    import torch
    from cosmoprimo_amd import _lib, _device as dv
    dev = torch.device('cuda', 0)
    rng = np.random.default_rng(11)
    mf, ms, off = 20, 5, (-10, 20)
    for n in (64, 257, 2048):
        dd = rng.standard_normal((301, n))
        dd[1, mf + 3] = dd[1, mf + 9] = 50.                  # tie: the first wins
        dd[2, n - mf - 1] = 60.                              # maximum on the last admissible index: second range empty
        dd[3, mf + 7] = np.nan                               # NaN is the maximum
        dd[4, :] = 1.                                        # constant row
        dd[5, 0] = dd[5, n - 1] = 1e9                        # outside the ranges: ignored
        dd[6, mf + 1] = 70.; dd[6, mf + 1 + ms] = 65.; dd[6, mf + ms] = 66.     # the second range starts at argmax + margin_second
        t = torch.as_tensor(dd, device=dev)
        box = torch.empty((dd.shape[0], 2), dtype=torch.int32, device=dev)
        _lib.check(_lib.load().cp_wallish_box(t.data_ptr(), dd.shape[0], n, mf, ms, off[0], off[1], box.data_ptr(), 0, dv.stream_of(dev)))
        ref = np.empty((dd.shape[0], 2), dtype=np.int64)
        for i, row in enumerate(dd):
            first = row[mf:n - mf].argmax() + mf
            tail = row[first + ms:n - mf]
            second = first + ms + tail.argmax() if tail.size else 0
            ref[i] = first + off[0], second + off[1]
        assert np.array_equal(box.cpu().numpy(), ref), n


@pytest.mark.parametrize('engine', ['eisenstein_hu', 'eisenstein_hu_nowiggle', 'bbks'])
def test_wallish_tail_in_one_kernel(cp, engine):
    """cp_wallish_tail (second derivatives + box + removal, inverse transform, spliced spline + damping of wallish2018 as one kernel, the transformed rows
    kept on the CU) and cp_wallish_full (the forward transform with its spectra in the same kernel) against the three calls they replace: pknow, the boxes and the rewritten sequences; odd batches (a vector without a partner) and a
    vector that is not finite next to good ones (it comes out as the three calls leave it, its partner untouched)."""
    import torch
    from cosmoprimo_amd import bao_filter as bf
    from cosmoprimo_amd.bao_filter import PowerSpectrumBAOFilter
    fid = cp.Cosmology(engine='eisenstein_hu')
    rng = np.random.default_rng(17)
    for nb in (129, 200, 513) if engine == 'eisenstein_hu' else (131,):
        par = dict(Omega_m=rng.uniform(0.24, 0.40, nb), Omega_b=rng.uniform(0.04, 0.06, nb), h=rng.uniform(0.6, 0.8, nb), n_s=rng.uniform(0.92, 1., nb))
        if nb == 200:
            par['h'][7] = np.nan      # a vector of NaN in the middle of a pair
        cosmo = cp.Cosmology(engine=engine, sigma8=0.8, **par)
        interp = cosmo.get_fourier().pk_interpolator(z=np.array([0.]))
        saved = bf._TAIL_IN_ONE_KERNEL, bf._ALL_IN_ONE_KERNEL
        try:
            got = []
            for fused, whole in ((True, False), (False, False), (True, True)):
                bf._TAIL_IN_ONE_KERNEL, bf._ALL_IN_ONE_KERNEL = fused, whole
                f = PowerSpectrumBAOFilter(interp, engine='wallish2018', cosmo=cosmo, cosmo_fid=fid)
                if whole:      # cp_wallish_full: the sequences are not written
                    assert f._even_now is None and f._odd_now is None
                    pk2, box2 = np.asarray(f.pknow), [b.cpu().numpy() for b in f._boxes]
                else:
                    got.append((np.asarray(f.pknow), [b.cpu().numpy() for b in f._boxes], f._even_now.cpu().numpy(), f._odd_now.cpu().numpy()))
        finally:
            bf._TAIL_IN_ONE_KERNEL, bf._ALL_IN_ONE_KERNEL = saved
        (pk1, box1, even1, odd1), (pk0, box0, even0, odd0) = got
        assert pk1.shape == pk0.shape == (nb, 1024, 1)
        good = np.isfinite(par['h'])
        assert np.isfinite(pk1[good]).all() and np.isnan(pk1[~good]).all() == np.isnan(pk0[~good]).all()
        np.testing.assert_allclose(pk1[good], pk0[good], rtol=1e-12)
        for a, b in zip(box1, box0):
            assert np.array_equal(a[good], b[good])
        np.testing.assert_allclose(even1[good], even0[good], rtol=1e-13, atol=1e-300)
        np.testing.assert_allclose(odd1[good], odd0[good], rtol=1e-13, atol=1e-300)
        # ... and the whole filter as one kernel
        assert np.isfinite(pk2[good]).all() and np.isnan(pk2[~good]).all() == np.isnan(pk0[~good]).all()
        np.testing.assert_allclose(pk2[good], pk0[good], rtol=1e-12)
        for a, b in zip(box2, box0):
            assert np.array_equal(a[good], b[good])
        if nb == 129:      # cp_wallish_full with d_coef: the rewritten sequences it can hand back are the ones the separate calls leave
            from cosmoprimo_amd import _lib, _device as dv
            from cosmoprimo_amd.background import DEFAULTS as bg_defaults
            from cosmoprimo_amd.power import PK_DEFAULTS
            ops = f._operators()
            name, bg, pk = interp._interp.analytic_engine()
            cbg, _, keep1 = dv.pack_params(_lib.BG_PARAMS, bg, bg_defaults, f.device)
            cpk, _, keep2 = dv.pack_params(_lib.PK_PARAMS, pk, PK_DEFAULTS, f.device)
            lib = _lib.load()
            rows = f._pk_rows.contiguous()
            coef = torch.empty((nb, 4096), dtype=torch.float64, device=f.device)
            box = torch.empty((2 * nb, 2), dtype=torch.int32, device=f.device)
            out = torch.empty_like(rows)
            work = torch.empty(int(lib.cp_dst_forward_analytic_workspace_bytes(nb)), dtype=torch.uint8, device=f.device)
            _lib.check(lib.cp_wallish_full(ops['dst']._handle, ops['splice']._handle, _lib.ENGINES[name], nb, dv.as_void_p(cbg), 0, None, dv.as_void_p(cpk), rows.data_ptr(),
                                           rows.shape[1], 20, 5, -10, 20, ops['tophat'].data_ptr(), box.data_ptr(), coef.data_ptr(), out.data_ptr(), work.data_ptr(),
                                           dv.stream_of(f.device)))
            seqs = coef.view(2 * nb, 2048).cpu().numpy()
            np.testing.assert_allclose(seqs[0::2], even0, rtol=1e-13, atol=1e-300)
            np.testing.assert_allclose(seqs[1::2], odd0, rtol=1e-13, atol=1e-300)
            np.testing.assert_allclose(out.cpu().numpy(), pk0[:, :, 0], rtol=1e-12)


def test_brieden2022_on_a_batch_of_tables(cp, golden):
    """brieden2022 with ONE cosmology on a batched 2D interpolator (a batch of (k, z) tables): every table as on its own."""
    from cosmoprimo_amd.bao_filter import PowerSpectrumBAOFilter
    g = golden('bao')
    fid = cp.Cosmology(engine='eisenstein_hu')
    cosmo = cp.Cosmology(engine='eisenstein_hu', **BAO_PARAMS[3])
    tables = np.stack([g['tab_pk'], 1.1 * g['tab_pk'][:, ::-1], g['tab_pk']**1.01])      # (3, nk, nz)
    batch = cp.PowerSpectrumInterpolator2D(g['tab_k'], g['tab_z'], tables)
    b = PowerSpectrumBAOFilter(batch, engine='brieden2022', cosmo=cosmo, cosmo_fid=fid)
    assert b.pknow.shape == (3, 1024, 4)
    np.testing.assert_allclose(b.pknow[0], g['tab_brieden_pknow'], rtol=RTOL)
    for i in range(3):
        one = PowerSpectrumBAOFilter(cp.PowerSpectrumInterpolator2D(g['tab_k'], g['tab_z'], tables[i]), engine='brieden2022', cosmo=cosmo, cosmo_fid=fid)
        np.testing.assert_allclose(b.pknow[i], one.pknow, rtol=1e-10)
