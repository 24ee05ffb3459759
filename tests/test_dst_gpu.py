"""GPU: batched orthonormal DST-II / DST-III against scipy.fftpack (the reference's calls, bao_filter.py:371-372, 412)."""
import numpy as np
import pytest
from scipy import fftpack

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('n', [256, 1024, 4096])
def test_dst_roundtrip_and_scipy(n):
    import torch
    from cosmoprimo_amd.dst import DST
    rng = np.random.default_rng(n)
    x = rng.normal(size=(7, n)) * np.linspace(1., 3., n)      # odd number of rows: incomplete last pair
    k = np.linspace(1e-7, 2., n)
    plan = DST(n, kx=k)
    y = plan(x).cpu().numpy()
    ref = fftpack.dst(x, type=2, axis=-1, norm='ortho')
    assert np.abs(y - ref).max() < 1e-13 * np.abs(ref).max()
    back = plan(torch.as_tensor(ref, device='cuda'), inverse=True).cpu().numpy()
    assert np.abs(back - fftpack.idst(ref, type=2, axis=-1, norm='ortho')).max() < 1e-13 * np.abs(x).max()
    assert np.abs(back - x).max() < 1e-13 * np.abs(x).max()
    # fused maps of the wallish filter: dst(log(k p)) and exp(idst(.)) / k
    p = np.exp(rng.normal(size=(4, n)) * 0.1) / k
    yf = plan(p, fused=True).cpu().numpy()
    reff = fftpack.dst(np.log(k * p), type=2, axis=-1, norm='ortho')
    assert np.abs(yf - reff).max() < 1e-13 * np.abs(reff).max()
    pb = plan(reff, inverse=True, fused=True).cpu().numpy()
    assert np.abs(pb / p - 1).max() < 1e-12
    # de-interleaved coefficient layout: [Y_0, Y_2, ... | Y_1, Y_3, ...], the same numbers
    ys = plan(x, split=True).cpu().numpy()
    assert np.array_equal(ys[:, :n // 2], y[:, 0::2]) and np.array_equal(ys[:, n // 2:], y[:, 1::2])
    assert np.array_equal(plan(torch.as_tensor(ys, device='cuda'), inverse=True, split=True).cpu().numpy(), plan(torch.as_tensor(y, device='cuda'), inverse=True).cpu().numpy())
    yfs = plan(p, fused=True, split=True).cpu().numpy()
    assert np.array_equal(yfs[:, :n // 2], yf[:, 0::2]) and np.array_equal(yfs[:, n // 2:], yf[:, 1::2])


def test_nonfinite_rows_stay_isolated():
    """Rows are transformed in pairs: a NaN row (or a non-positive one under the fused log map) must not reach its partner."""
    import torch
    from cosmoprimo_amd.dst import DST
    from scipy import fftpack
    rng = np.random.default_rng(5)
    x = rng.uniform(0.5, 2., (4, 1024))
    x[1, 3] = np.nan
    d = DST(1024)
    out = d(torch.as_tensor(x, device='cuda')).cpu().numpy()
    assert np.isnan(out[1]).all()
    for i in (0, 2, 3):
        np.testing.assert_allclose(out[i], fftpack.dst(x[i], type=2, norm='ortho'), rtol=1e-11, atol=1e-12)
    kx = np.linspace(1e-3, 2., 1024)
    x[1, 3], x[2, 10] = 1., -1.
    out = DST(1024, kx=kx)(torch.as_tensor(x, device='cuda'), fused=True).cpu().numpy()
    assert np.isnan(out[2]).all()
    for i in (0, 1, 3):
        np.testing.assert_allclose(out[i], fftpack.dst(np.log(kx * x[i]), type=2, norm='ortho'), rtol=1e-10, atol=1e-11)


def test_rows_screen():
    """cp_rows_screen (the one-pass row check the FFTLog / DST facades run before a paired transform) against the torch expressions it replaced."""
    import torch
    from cosmoprimo_amd import _device as dv
    dev = torch.device('cuda', 0)
    rng = np.random.default_rng(5)
    for n in (1, 63, 256, 1000, 4096):
        x = rng.standard_normal((37, n)) * 10.**rng.uniform(-300, 300, (37, 1))
        x[3, n // 2] = np.nan
        x[5, 0] = np.inf
        x[7, n - 1] = -np.inf
        x[11] = 0.
        x[13] = np.abs(x[13]) + 1e-300
        x[17] = 2.**-1074                      # subnormal
        x[19] = np.nextafter(np.inf, 0.)       # largest finite value
        t = torch.as_tensor(x, device=dev)
        ok, scale = dv.screen_rows(t, with_scale=True)
        finite = torch.isfinite(t).all(dim=-1, keepdim=True)
        assert ok.shape == (37, 1) and torch.equal(ok, finite)
        assert torch.equal(dv.screen_rows(t, require_positive=True), finite & (t > 0.).all(dim=-1, keepdim=True))
        # 2^e >= max|row| with e from frexp, 1 for an all-zero row: an exact power of two (torch.ldexp, which the facade used before, goes
        # through pow() and is one ulp off for some exponents, so that the "rescaling by powers of two" it stood for was not exact)
        good = finite[:, 0].cpu().numpy()
        amax = np.abs(x[good]).max(axis=1)
        ref = np.ldexp(1., np.minimum(np.frexp(amax)[1], 1023))       # (2^1024 is not a double: the largest rows are scaled to < 2)
        assert np.array_equal(scale.cpu().numpy()[good, 0], ref)
        assert np.all(np.frexp(ref)[0] == 0.5) and np.all(ref >= amax / 2.) and np.all(ref[amax < 2.**1023] >= amax[amax < 2.**1023])
    assert dv.screen_rows(torch.empty((0, 8), dtype=torch.float64, device=dev)).shape == (0, 1)
    big = torch.as_tensor(rng.standard_normal((70000, 16)), device=dev)      # more rows than workgroups
    big[69999, 3] = float('nan')
    ok = dv.screen_rows(big)
    assert bool(ok[:69999].all()) and not bool(ok[69999])


@pytest.mark.parametrize('engine', ['eisenstein_hu', 'eisenstein_hu_nowiggle', 'bbks'])
def test_forward_transform_that_evaluates_the_spectra(engine):
    """cp_dst_forward_analytic (log(k P_c(k)) of a batch of analytic cosmologies evaluated inside the transform kernel) against the two launches it
    replaces (cp_power_eval CP_PK_LOG_K_MATTER, then cp_dst_execute) and, through them, scipy: odd batches (a cosmology without a partner), both
    coefficient layouts, a cosmology whose parameters give NaN next to good ones."""
    import torch
    from cosmoprimo_amd import power
    from cosmoprimo_amd.dst import DST
    dev = torch.device('cuda', 0)
    klin = np.linspace(1e-4, 5., 4096)
    dst = DST(4096, kx=klin, device=dev)
    for n in (1, 2, 7, 130):
        rng = np.random.default_rng(n)
        Om, Ob, h, ns = rng.uniform(.25, .40, n), rng.uniform(.04, .06, n), rng.uniform(.6, .8, n), rng.uniform(.92, 1., n)
        if n == 7:
            ns[4] = np.nan
        bg = dict(h=torch.as_tensor(h, device=dev), Omega_cdm=torch.as_tensor(Om - Ob, device=dev), Omega_b=torch.as_tensor(Ob, device=dev))
        pk = dict(n_s=torch.as_tensor(ns, device=dev), A_s=torch.as_tensor(rng.uniform(1.8e-9, 2.4e-9, n), device=dev))
        rows = power.analytic(engine, 'log_k_matter', klin, bg=bg, pk=pk, device=dev).reshape(n, 4096)
        for split in (False, True):
            ref = dst(rows, split=split).cpu().numpy()
            got = dst.forward_analytic(engine, bg, pk, split=split)
            assert got is not None and got.shape == (n, 4096)
            got = got.cpu().numpy()
            assert np.array_equal(np.isnan(got), np.isnan(ref))
            if n == 7:
                assert np.isnan(got[4]).all() and np.isfinite(np.delete(got, 4, axis=0)).all()
            scale = np.abs(ref[np.isfinite(ref).all(axis=1)]).max(axis=1, keepdims=True)
            assert np.abs((got - ref)[np.isfinite(ref).all(axis=1)] / scale).max() < 1e-14
        host = fftpack.dst(rows[:1].cpu().numpy(), type=2, axis=-1, norm='ortho')
        np.testing.assert_allclose(dst.forward_analytic(engine, bg, pk)[:1].cpu().numpy(), host, rtol=0, atol=1e-12 * np.abs(host).max())
    assert DST(1024, kx=np.linspace(1e-4, 5., 1024), device=dev).forward_analytic(engine, bg, pk) is None


@pytest.mark.parametrize('engine', ['eisenstein_hu', 'eisenstein_hu_nowiggle'])
def test_forward_transform_that_also_finds_and_rewrites_the_boxes(engine):
    """cp_dst_forward_analytic_box against the two calls it replaces (cp_dst_forward_analytic with the split layout, then cp_wallish_dd_box in
    place): same coefficients to the last bit outside the boxes, same boxes, the rewritten knots to 1e-13 of the sequence's scale; odd batches
    (a cosmology without a partner), a NaN cosmology next to good ones."""
    import torch
    from cosmoprimo_amd import _lib, _device as dv
    from cosmoprimo_amd.dst import DST
    dev = torch.device('cuda', 0)
    klin = np.linspace(1e-4, 5., 4096)
    dst = DST(4096, kx=klin, device=dev)
    lib = _lib.load()
    mf, ms, off = 20, 5, (-10, 20)      # the filter's own margins and offsets (bao_filter.py:388-394)
    for n in (1, 2, 7, 301):
        rng = np.random.default_rng(n)
        Om, Ob, h, ns = rng.uniform(.25, .40, n), rng.uniform(.04, .06, n), rng.uniform(.6, .8, n), rng.uniform(.92, 1., n)
        if n == 7:
            ns[4] = np.nan
        bg = dict(h=torch.as_tensor(h, device=dev), Omega_cdm=torch.as_tensor(Om - Ob, device=dev), Omega_b=torch.as_tensor(Ob, device=dev))
        pk = dict(n_s=torch.as_tensor(ns, device=dev), A_s=torch.as_tensor(rng.uniform(1.8e-9, 2.4e-9, n), device=dev))
        plain = dst.forward_analytic(engine, bg, pk, split=True)
        y = plain.clone().view(2 * n, 2048)
        box_ref = torch.empty((2 * n, 2), dtype=torch.int32, device=dev)
        _lib.check(lib.cp_wallish_dd_box(y.data_ptr(), 2 * n, 2048, mf, ms, off[0], off[1], box_ref.data_ptr(), None, y.data_ptr(), 0, dv.stream_of(dev)))
        got, box = dst.forward_analytic(engine, bg, pk, split=True, box=(mf, ms, off[0], off[1]))
        got, box, y, box_ref, plain = (v.cpu().numpy() for v in (got.view(2 * n, 2048), box, y, box_ref, plain.view(2 * n, 2048)))
        good = np.isfinite(plain).all(axis=1)
        assert good.sum() == 2 * n - (2 if n == 7 else 0)
        assert np.array_equal(box[good], box_ref[good])
        assert np.array_equal(np.isnan(got), np.isnan(y))
        for i in np.flatnonzero(good):
            a, b = box[i]
            inside = np.zeros(2048, dtype=bool)
            if a >= 1 and b <= 2046 and b >= a:
                inside[a:b + 1] = True
                assert not np.array_equal(got[i, inside], plain[i, inside])          # the box was rewritten
            assert np.array_equal(got[i, ~inside], plain[i, ~inside])
            assert np.abs(got[i, inside] - y[i, inside]).max(initial=0.) <= 1e-13 * np.abs(plain[i]).max()
