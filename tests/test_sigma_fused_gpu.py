"""GPU: the fused sigma(r, z) kernel (csrc/cp_sigma.hip, cp_sigma_rz_analytic with nblocks = 0) against the three separate kernels it replaces and
against the oracle: every analytic engine, odd batches, odd numbers of radii / redshifts, a cosmology whose parameters give NaN next to good
ones, the two-stream block walk of the same entry point."""
import warnings

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def parameters(n, seed):
    rng = np.random.default_rng(seed)
    return dict(Omega_m=rng.uniform(.25, .40, n), Omega_b=rng.uniform(.04, .06, n), h=rng.uniform(.6, .8, n), n_s=rng.uniform(.92, 1., n))


def both_routes(interp, r, z):
    kind = type(interp)
    saved = kind._two_stream_min_bytes
    try:
        kind._two_stream_min_bytes = 0
        fused = interp.sigma_rz(r, z)
        kind._two_stream_min_bytes = 1 << 60
        separate = interp.sigma_rz(r, z)
    finally:
        kind._two_stream_min_bytes = saved
    return fused, separate


@pytest.mark.parametrize('prefiltered', [False, True])
@pytest.mark.parametrize('engine', ['eisenstein_hu', 'eisenstein_hu_nowiggle', 'bbks'])
def test_fused_matches_separate_kernels(engine, prefiltered, monkeypatch):
    """prefiltered: the spline of the fused kernel evaluated from the B-spline coefficients its transform delivers (cp_sigma_rz_analytic_prefiltered: the
    transform runs with another u, so the comparison is that of two evaluations of an FFTLog -- 1e-14 of the tilted transform's scale, a few 1e-13 on sigma at
    these radii) instead of multiplied by the banded operator (the same arithmetic as the separate kernels: 1e-13)."""
    import cosmoprimo_amd as cp
    from cosmoprimo_amd import interpolator as itp
    warnings.simplefilter('ignore')
    monkeypatch.setattr(itp, '_SIGMA_RZ_PREFILTERED', prefiltered)
    for key in [key for key in itp._op_cache if key[0] == 'geospline']:
        del itp._op_cache[key]
    for n, nr, nz in ((1, 5, 3), (7, 256, 64), (64, 130, 7), (301, 33, 16)):
        interp = cp.Cosmology(engine=engine, sigma8=0.8, **parameters(n, n)).get_fourier().pk_interpolator()
        r, z = np.geomspace(0.5, 150., nr), np.linspace(0., 2.5, nz)
        fused, separate = both_routes(interp, r, z)
        assert fused.shape == (n, nr, nz) and np.isfinite(fused).all()
        np.testing.assert_allclose(fused, separate, rtol=2e-12 if prefiltered else 1e-13, atol=0, err_msg=str((engine, n, nr, nz)))
    plans = [v for key, v in itp._op_cache.items() if key[0] == 'geospline']
    assert bool(plans) == prefiltered and all(p.prefiltered for p in plans)      # (the route taken is the one asked for)


def test_fused_against_oracle():
    import cosmoprimo_amd as cp
    from oracle import background as ob, power as op, sigma as osig
    warnings.simplefilter('ignore')
    par = parameters(5, 11)
    r, z = np.geomspace(1., 100., 40), np.linspace(0., 3., 6)
    interp = cp.Cosmology(engine='eisenstein_hu', sigma8=0.8, **par).get_fourier().pk_interpolator()
    fused, _ = both_routes(interp, r, z)
    for i in range(5):
        Om, Ob, h, ns = (float(par[name][i]) for name in ('Omega_m', 'Omega_b', 'h', 'n_s'))
        g2 = op.growth_factor(z, ob.derived(h=h, Omega_b=Ob, Omega_m=Om), znorm=0.)**2
        pk0 = lambda k: op.pk_z0(k, 'eisenstein_hu', h=h, Omega_cdm=Om - Ob, Omega_b=Ob, n_s=ns)        # noqa: E731
        norm = 0.8**2 / (float(osig.sigma_r2(np.array([8.]), pk0)[0]) * g2[0])
        ref = (norm * osig.sigma_r2(r, lambda k: pk0(k)[:, None] * g2[None, :]))**0.5
        np.testing.assert_allclose(fused[i], ref, rtol=1e-9)


def test_rows_stay_independent_and_radii_outside_are_nan():
    import torch
    import cosmoprimo_amd as cp
    warnings.simplefilter('ignore')
    par = parameters(6, 5)
    good = cp.Cosmology(engine='eisenstein_hu', **par).get_fourier().pk_interpolator()
    r, z = np.geomspace(1., 100., 24), np.linspace(0., 1., 4)
    ref, _ = both_routes(good, r, z)
    bad = {name: torch.as_tensor(v, device='cuda') for name, v in par.items()}
    bad['n_s'] = bad['n_s'].clone()
    bad['n_s'][2] = float('nan')                   # cosmology 2 shares a transform with cosmology 3
    mixed = cp.Cosmology(engine='eisenstein_hu', **bad).get_fourier().pk_interpolator()
    fused, separate = both_routes(mixed, torch.as_tensor(r, device='cuda'), torch.as_tensor(z, device='cuda'))
    fused, separate = fused.cpu().numpy(), separate.cpu().numpy()
    assert np.isnan(fused[2]).all() and np.isnan(separate[2]).all()
    keep = [0, 1, 3, 4, 5]
    np.testing.assert_allclose(fused[keep], ref[keep], rtol=1e-12)
    # radii outside the transform's output grid: NaN, as the spline returns them
    wide = np.array([1e-3, 1., 10., 1e8])
    fused, separate = both_routes(good, wide, z)
    assert np.array_equal(np.isnan(fused), np.isnan(separate)) and np.isfinite(fused[:, 1:3]).all()


def test_block_walk_on_two_streams():
    """cp_sigma_rz_analytic with nblocks > 0: the three separate kernels, blocks of cosmologies, the store on a second stream."""
    import cosmoprimo_amd as cp
    warnings.simplefilter('ignore')
    interp = cp.Cosmology(engine='eisenstein_hu', **parameters(203, 9)).get_fourier().pk_interpolator()
    r, z = np.geomspace(1., 100., 64), np.linspace(0., 2., 9)
    kind = type(interp)
    fused, separate = both_routes(interp, r, z)
    saved = kind._two_stream_min_bytes, kind._two_stream_blocks
    try:
        kind._two_stream_min_bytes = 0
        for blocks in (1, 3, 16):
            kind._two_stream_blocks = blocks
            np.testing.assert_allclose(interp.sigma_rz(r, z), separate, rtol=1e-13, atol=0)      # (pairs of cosmologies differ at block edges)
    finally:
        kind._two_stream_min_bytes, kind._two_stream_blocks = saved


@pytest.mark.parametrize('engine', ['eisenstein_hu', 'eisenstein_hu_nowiggle', 'bbks'])
def test_few_radii_as_a_functional_of_the_spectrum(engine):
    """cp_sigma_rz_functional (up to four radii: the sigma8 normalisation) against the fused kernel of the same helper: sigma(r, z), and the spectra
    it hands back, which are cp_power_eval's to a few ulp; odd batches, one to four radii, a NaN cosmology next to good ones."""
    import torch
    from cosmoprimo_amd import interpolator as itp
    warnings.simplefilter('ignore')
    dev = torch.device('cuda', torch.cuda.current_device())
    for n, nr, nz in ((1, 1, 1), (7, 1, 1), (130, 3, 5), (33, 4, 2)):
        par = parameters(n, 3 * n + nr)
        if n == 7:
            par['n_s'][4] = np.nan
        bg = dict(h=torch.as_tensor(par['h'], device=dev), Omega_cdm=torch.as_tensor(par['Omega_m'] - par['Omega_b'], device=dev),
                  Omega_b=torch.as_tensor(par['Omega_b'], device=dev))
        pk = dict(n_s=torch.as_tensor(par['n_s'], device=dev))
        r = np.geomspace(2., 60., nr) if nr > 1 else np.array([8.])
        g2 = torch.as_tensor(np.random.default_rng(n).uniform(0.2, 1., (n, nz)), device=dev)
        saved = itp._FUNCTIONAL_RADII
        try:
            itp._FUNCTIONAL_RADII = 4
            out_f, pk_f, k = itp.sigma_rz_analytic(engine, bg, pk, r, g2, dev, keep_spectra=True)
            itp._FUNCTIONAL_RADII = 0
            out_t, pk_t, _ = itp.sigma_rz_analytic(engine, bg, pk, r, g2, dev, keep_spectra=True)
        finally:
            itp._FUNCTIONAL_RADII = saved
        out_f, out_t, pk_f, pk_t = (t.cpu().numpy() for t in (out_f, out_t, pk_f, pk_t))
        assert out_f.shape == (n, nr, nz)
        assert np.array_equal(np.isnan(out_f), np.isnan(out_t))
        if n == 7:
            assert np.isnan(out_f[4]).all() and np.isfinite(np.delete(out_f, 4, axis=0)).all()
        np.testing.assert_allclose(out_f, out_t, rtol=2e-13, atol=0, err_msg=str((engine, n, nr, nz)))
        # the fused kernel steps k geometrically (1e-15 on k after 8 steps of 128 samples); BBKS as coded takes log(1 + 2.34 q) / (2.34 q) at q ~ 1e-5,
        # where an ulp of q is 1e-11 of the ratio (bbks.py:64)
        np.testing.assert_allclose(pk_f, pk_t, rtol=1e-10 if engine == 'bbks' else 2e-13, atol=0)
        from cosmoprimo_amd import power
        direct = power.analytic(engine, 'matter', k, bg=bg, pk=pk, device=dev).cpu().numpy().reshape(pk_f.shape)
        # the same evaluation but for log k (cp_power_eval takes it per sample by the table-driven form, this kernel reads the launch's table) and the
        # two powers of k the EH98 fit takes: cp_power_eval forms them as exp(p (log k + log c)), this kernel as k^p c^p with k^p tabulated once per
        # launch (cp_power_eval.h: transfer_eh_powers) -- a few ulp apart
        assert np.array_equal(np.isnan(direct), np.isnan(pk_f))
        np.testing.assert_allclose(pk_f, direct, rtol=4e-15, atol=0)


def _oracle_sigma(engine, par, i, r, g2):
    """sqrt(sigma^2(r) growth_sq) of cosmology i straight from the oracle (oracle/sigma.py: integrate_sigma_r2 restated, interpolator.py:200-292),
    with the fiducial amplitude the entry points default to."""
    from oracle import power as op, sigma as osig
    Om, Ob, h, ns = (float(par[name][i]) for name in ('Omega_m', 'Omega_b', 'h', 'n_s'))
    pk0 = lambda k: op.pk_z0(k, engine, h=h, Omega_cdm=Om - Ob, Omega_b=Ob, n_s=ns, sigma8=0.8)        # noqa: E731
    k = np.geomspace(1e-7, 1e2, 1024)
    return (osig.sigma_r2(r, pk0)[:, None] * g2[None, :])**0.5, pk0(k)


@pytest.mark.parametrize('engine', ['eisenstein_hu', 'eisenstein_hu_nowiggle', 'bbks'])
def test_functional_against_the_oracle(engine):
    """cp_sigma_rz_functional pinned on the oracle itself (not on the fused kernel): sigma8 and sigma(r) at up to four radii of sampled
    cosmologies, and the spectra it hands back, 1e-10."""
    import torch
    from cosmoprimo_amd import interpolator as itp
    warnings.simplefilter('ignore')
    dev = torch.device('cuda', torch.cuda.current_device())
    for n, radii in ((5, [8.]), (9, [2., 8., 25., 90.]), (3, [0.7, 140.])):
        par = parameters(n, 40 + n)
        bg = dict(h=torch.as_tensor(par['h'], device=dev), Omega_cdm=torch.as_tensor(par['Omega_m'] - par['Omega_b'], device=dev),
                  Omega_b=torch.as_tensor(par['Omega_b'], device=dev))
        pk = dict(n_s=torch.as_tensor(par['n_s'], device=dev))
        g2 = np.random.default_rng(n).uniform(0.2, 1., (n, 3))
        r = np.array(radii)
        assert r.size <= itp._FUNCTIONAL_RADII
        out, spectra, k = itp.sigma_rz_analytic(engine, bg, pk, r, torch.as_tensor(g2, device=dev), dev, keep_spectra=True)
        out, spectra = out.cpu().numpy(), spectra.cpu().numpy()
        for i in range(n):
            ref, pk_ref = _oracle_sigma(engine, par, i, r, g2[i])
            np.testing.assert_allclose(out[i], ref, rtol=1e-10, atol=0, err_msg=str((engine, n, i)))
            np.testing.assert_allclose(spectra[i], pk_ref, rtol=1e-10, atol=0)


@pytest.mark.parametrize('blocks', [1, 2, 5])
def test_block_route_keeps_spectra_inside_their_buffer(blocks):
    """cp_sigma_rz_analytic, block route, with d_pk_out given: the (ncosmo, nk) spectra buffer is exactly that size (variances and coefficients
    belong to the workspace); a guard region behind the spectra must stay untouched and the results must be the fused route's."""
    import torch
    from cosmoprimo_amd import _lib, _device as dv, interpolator as itp
    from cosmoprimo_amd.background import DEFAULTS as bg_defaults
    from cosmoprimo_amd.power import PK_DEFAULTS
    warnings.simplefilter('ignore')
    dev = torch.device('cuda', torch.cuda.current_device())
    n, nz, nk = 37, 3, 1024
    par = parameters(n, 77)
    bg = dict(h=torch.as_tensor(par['h'], device=dev), Omega_cdm=torch.as_tensor(par['Omega_m'] - par['Omega_b'], device=dev),
              Omega_b=torch.as_tensor(par['Omega_b'], device=dev))
    pk = dict(n_s=torch.as_tensor(par['n_s'], device=dev))
    r = np.geomspace(1., 100., 48)
    g2 = torch.as_tensor(np.random.default_rng(3).uniform(0.2, 1., (n, nz)), device=dev)
    ref, pk_ref, _ = itp.sigma_rz_analytic('eisenstein_hu', bg, pk, r, g2, dev, keep_spectra=True)
    got, pk_got, _ = itp.sigma_rz_analytic('eisenstein_hu', bg, pk, r, g2, dev, blocks=blocks, keep_spectra=True)
    np.testing.assert_allclose(got.cpu().numpy(), ref.cpu().numpy(), rtol=1e-13, atol=0)
    np.testing.assert_allclose(pk_got.cpu().numpy(), pk_ref.cpu().numpy(), rtol=2e-13, atol=0)
    # the same call through the C ABI with the spectra at the head of a larger allocation whose tail is a canary
    lib = _lib.load()
    cbg, _, keep1 = dv.pack_params(_lib.BG_PARAMS, bg, bg_defaults, dev)
    cpk, _, keep2 = dv.pack_params(_lib.PK_PARAMS, pk, PK_DEFAULTS, dev)
    fft = itp.TophatVariance(np.geomspace(1e-7, 1e2, nk), device=dev)
    op = itp.LinearOperator.spline(fft.y[0], r, bc='natural', device=dev)
    guard = 3 * n * nk
    buf = torch.full((n * nk + guard,), -7.25, dtype=torch.float64, device=dev)
    out = torch.empty((n, r.size, nz), dtype=torch.float64, device=dev)
    work = torch.empty(int(lib.cp_sigma_rz_workspace_bytes(n, nk)), dtype=torch.uint8, device=dev)
    kdev = dv.upload(np.geomspace(1e-7, 1e2, nk), dev)
    _lib.check(lib.cp_sigma_rz_analytic(_lib.ENGINES['eisenstein_hu'], n, dv.as_void_p(cbg), 0, None, dv.as_void_p(cpk), nk, kdev.data_ptr(),
                                        fft._get_plan(dev).handle, op._handle, g2.data_ptr(), nz, out.data_ptr(), buf.data_ptr(), work.data_ptr(),
                                        blocks, dev.index, dv.stream_of(dev)))
    torch.cuda.synchronize()
    assert bool((buf[n * nk:] == -7.25).all()), 'cp_sigma_rz_analytic wrote behind the (ncosmo, nk) spectra it was given'
    np.testing.assert_allclose(out.cpu().numpy(), ref.cpu().numpy(), rtol=1e-13, atol=0)


@pytest.mark.parametrize('engine', ['eisenstein_hu', 'eisenstein_hu_nowiggle', 'bbks'])
def test_sigma8_normalisation_kernel_against_the_oracle(engine):
    """cp_sigma8_normalise (factors, normalised amplitudes and normalised spectra of a batch in one kernel) against the oracle's restatement of the
    reference's normalisation (eisenstein_hu.py:94-103: sigma8 of the first-guess amplitude by the FFTLog route, growth factor at z = 0 included),
    1e-10; one target for all cosmologies and one per cosmology; a NaN cosmology next to good ones."""
    import torch
    from cosmoprimo_amd import interpolator as itp
    from oracle import background as ob, power as op, sigma as osig
    warnings.simplefilter('ignore')
    dev = torch.device('cuda', torch.cuda.current_device())
    n = 9
    par = parameters(n, 123)
    par['n_s'][6] = np.nan
    bg = dict(h=torch.as_tensor(par['h'], device=dev), Omega_cdm=torch.as_tensor(par['Omega_m'] - par['Omega_b'], device=dev),
              Omega_b=torch.as_tensor(par['Omega_b'], device=dev))
    fid = 2.43e-9 * (0.8 / 0.87659)**2
    targets = np.random.default_rng(5).uniform(0.7, 0.9, n)
    for target in (0.8, torch.as_tensor(targets, device=dev), targets):
        pk = dict(n_s=torch.as_tensor(par['n_s'], device=dev), A_s=fid)
        rs, amp, spectra, k = itp.sigma8_normalise(engine, bg, pk, target, dev)
        rs, amp, spectra = rs.cpu().numpy(), amp.cpu().numpy(), spectra.cpu().numpy()
        assert np.isnan(rs[6]) and np.isnan(spectra[6]).all() and np.isfinite(np.delete(rs, 6)).all()
        want = np.broadcast_to(np.asarray(target.cpu() if torch.is_tensor(target) else target, dtype='f8'), (n,))
        for i in (0, 3, 8):
            Om, Ob, h, ns = (float(par[name][i]) for name in ('Omega_m', 'Omega_b', 'h', 'n_s'))
            g0 = float(op.growth_factor(np.zeros(1), ob.derived(h=h, Omega_b=Ob, Omega_m=Om), znorm=0.)[0])
            pk0 = lambda kk: op.pk_z0(kk, engine, h=h, Omega_cdm=Om - Ob, Omega_b=Ob, n_s=ns, A_s=fid)        # noqa: E731
            sigma8_fid = float(osig.sigma_r2(np.array([8.]), pk0)[0])**0.5 * g0
            np.testing.assert_allclose(rs[i], want[i] / sigma8_fid, rtol=1e-10)
            np.testing.assert_allclose(amp[i], fid * (want[i] / sigma8_fid)**2, rtol=1e-10)
            np.testing.assert_allclose(spectra[i], pk0(k) * (want[i] / sigma8_fid)**2, rtol=1e-10)


def test_batch_normalised_by_the_kernel_equals_one_cosmology_at_a_time():
    """Cosmology(sigma8=..., batch on the device) takes cp_sigma8_normalise; every entry must be what the same cosmology gives on its own (the
    scalar path: separate kernels), and sigma8_z(0) must come back as the target."""
    import torch
    import cosmoprimo_amd as cp
    warnings.simplefilter('ignore')
    par = parameters(5, 31)
    dev = torch.device('cuda', torch.cuda.current_device())
    batch = cp.Cosmology(engine='eisenstein_hu', sigma8=0.83, **{name: torch.as_tensor(v, device=dev) for name, v in par.items()})
    fo = batch.get_fourier()
    assert '_pk0_normalised' in batch.engine.__dict__      # the kernel's route was taken
    k = np.geomspace(1e-3, 1., 50)
    pk_batch = fo.pk_interpolator()(k, z=0.5)
    np.testing.assert_allclose(cp.interpolator._host(fo.sigma8_z(0.)), 0.83, rtol=1e-10)
    for i in range(5):
        one = cp.Cosmology(engine='eisenstein_hu', sigma8=0.83, **{name: float(v[i]) for name, v in par.items()})
        np.testing.assert_allclose(cp.interpolator._host(pk_batch)[i], one.get_fourier().pk_interpolator()(k, z=0.5), rtol=1e-11)
        np.testing.assert_allclose(cp.interpolator._host(batch.get_primordial().A_s)[i], one.get_primordial().A_s, rtol=1e-11)
