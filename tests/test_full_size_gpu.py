"""BASELINE.json configs 3, 4 and 5 at the sizes one GPU gets: sampled units against the oracle (oracle/, the numpy restatement of the
reference) and size-independent properties -- bitwise determinism, independence of the batch a unit sits in, monotonicity, idempotence of the
smoothing.  (Config 2 at full size: tests/test_fftlog_gpu.py.)"""
import os
import sys
import warnings

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_config3_sigma_rz_full_size():
    import torch
    import cosmoprimo_amd as cp
    from oracle import background as ob, power as op, sigma as osig
    warnings.simplefilter('ignore')
    nb = 10000
    rng = np.random.default_rng(1)
    par = dict(Omega_m=rng.uniform(.25, .40, nb), Omega_b=rng.uniform(.04, .06, nb), h=rng.uniform(.6, .8, nb), n_s=rng.uniform(.92, 1., nb))
    r, z = np.geomspace(1, 100, 256), np.linspace(0, 3, 64)
    interp = cp.Cosmology(engine='eisenstein_hu', sigma8=0.8, **par).get_fourier().pk_interpolator()
    out = interp.sigma_rz(r, z)
    assert out.shape == (nb, 256, 64) and np.all(np.isfinite(out))
    assert np.array_equal(out, interp.sigma_rz(r, z))                                     # bitwise determinism
    assert np.all(np.diff(out, axis=1) < 0.) and np.all(np.diff(out, axis=2) < 0.)        # sigma falls with radius and with redshift
    i8 = np.argmin(np.abs(r - 8.))
    s8 = interp.sigma8_z(0.)
    assert np.allclose(s8, 0.8, rtol=1e-9) and abs(r[i8] - 8.) > 1e-3                    # normalised at r = 8, which is not on the grid
    for i in (0, 4321, nb - 1):                                                            # the reference's own path for single cosmologies
        Om, Ob, h, ns = (float(par[name][i]) for name in ('Omega_m', 'Omega_b', 'h', 'n_s'))
        bg = ob.derived(h=h, Omega_b=Ob, Omega_m=Om)
        g2 = op.growth_factor(z, bg, znorm=0.)**2
        pk0 = lambda k: op.pk_z0(k, 'eisenstein_hu', h=h, Omega_cdm=Om - Ob, Omega_b=Ob, n_s=ns)        # noqa: E731
        norm = 0.8**2 / (float(osig.sigma_r2(np.array([8.]), pk0)[0]) * g2[0])       # sigma8 is set at z = 0, growth factor (not 1 there) included
        ref = (norm * osig.sigma_r2(r, lambda k: pk0(k)[:, None] * g2[None, :]))**0.5
        np.testing.assert_allclose(out[i], ref, rtol=1e-9)
    # the batch took the fused kernel (cp_sigma.hip: P(k) -> FFTLog -> spline -> store in one launch); the three separate kernels do the same
    # arithmetic in the same order
    kind = type(interp)
    assert nb * 256 * 64 * 8 >= kind._two_stream_min_bytes
    saved, kind._two_stream_min_bytes = kind._two_stream_min_bytes, 1 << 60
    try:
        np.testing.assert_allclose(out, interp.sigma_rz(r, z), rtol=1e-13, atol=0)
    finally:
        kind._two_stream_min_bytes = saved
    # a cosmology gives the same numbers whatever batch it sits in
    sub = slice(1000, 1003)
    small = cp.Cosmology(engine='eisenstein_hu', sigma8=0.8, **{name: v[sub] for name, v in par.items()}).get_fourier().pk_interpolator().sigma_rz(r, z)
    np.testing.assert_allclose(small, out[sub], rtol=1e-12)


def test_config4_filters_full_chunk():
    import torch
    import cosmoprimo_amd as cp
    from oracle import bao as obao, power as op
    warnings.simplefilter('ignore')
    nb = 16384
    rng = np.random.default_rng(2)
    par = dict(Omega_m=rng.uniform(.25, .40, nb), Omega_b=rng.uniform(.04, .06, nb), h=rng.uniform(.6, .8, nb), n_s=rng.uniform(.92, 1., nb))
    cosmo = cp.Cosmology(engine='eisenstein_hu', sigma8=0.8, **par)
    interp = cosmo.get_fourier().pk_interpolator(z=np.array([0.]))
    filt = cp.PowerSpectrumBAOFilter(interp, engine='wallish2018')
    pknow = filt.pknow[..., 0]                                                             # (nb, 1024)
    assert pknow.shape == (nb, 1024) and np.all(np.isfinite(pknow)) and np.all(pknow > 0.)
    assert np.array_equal(pknow, cp.PowerSpectrumBAOFilter(interp, engine='wallish2018').pknow[..., 0])
    wiggles = filt.pk[..., 0] / pknow
    # BAO are a < 20 % feature; below k = 5e-4 the input is kept as it is (bao_filter.py:417)
    assert np.all(np.abs(wiggles - 1.) < 0.2) and np.all(np.abs(wiggles[:, filt.k < 4e-4] - 1.) < 1e-9)
    rsig = cp.interpolator._host(cosmo._engine._rsigma8)
    for i in (0, 7777, nb - 1):
        Om, Ob, h, ns = (float(par[name][i]) for name in ('Omega_m', 'Omega_b', 'h', 'n_s'))
        ref = obao.wallish2018(lambda k: op.pk_z0(k, 'eisenstein_hu', h=h, Omega_cdm=Om - Ob, Omega_b=Ob, n_s=ns, rsigma8=float(rsig[i]))[:, None])[:, 0]
        np.testing.assert_allclose(pknow[i], ref, rtol=1e-9)
    # smoothing what is already smooth changes little: the filter is (nearly) idempotent
    smooth = cp.PowerSpectrumInterpolator1D(filt.k, pknow[:64].T)
    again = cp.PowerSpectrumBAOFilter(smooth, engine='wallish2018').pknow
    mid = (filt.k > 1e-2) & (filt.k < 1.)
    assert np.all(np.abs(again.T[:, mid] / pknow[:64][:, mid] - 1.) < 2e-2)                # (1 % at k = 0.01: a smoother, not a projector)
    # brieden2022 on the same chunk: one rs_drag ratio per cosmology, rows of a small batch reproduce the rows of the large one
    fid = cp.Cosmology(engine='eisenstein_hu')
    big = cp.PowerSpectrumBAOFilter(interp, engine='brieden2022', cosmo=cosmo, cosmo_fid=fid).pknow[..., 0]
    assert big.shape == (nb, 1024) and np.all(np.isfinite(big))
    sub = slice(5000, 5003)
    csub = cp.Cosmology(engine='eisenstein_hu', sigma8=0.8, **{name: v[sub] for name, v in par.items()})
    small = cp.PowerSpectrumBAOFilter(csub.get_fourier().pk_interpolator(z=np.array([0.])), engine='brieden2022', cosmo=csub, cosmo_fid=fid).pknow[..., 0]
    np.testing.assert_allclose(small, big[sub], rtol=1e-10)
    bao = (filt.k > 0.05) & (filt.k < 0.5)
    dev = np.abs(big[:, bao] / pknow[:, bao] - 1.)                                         # two smoothing recipes: alike over the BAO range, not equal
    assert np.median(dev) < 0.02 and dev.max() < 0.2


def _benched_chunk():
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    return bench.CONFIG4_CHUNK


@pytest.mark.parametrize('chunk', sorted({16384, _benched_chunk()}))
def test_config4_one_gpu_share(chunk):
    """Config 4 at the size one of 8 GPUs gets: 125 000 EH98 P(k) vectors through wallish2018 and brieden2022, chunk by chunk as bench.py
    runs them (results kept on the device) -- at the chunk size bench.py times (bench.CONFIG4_CHUNK, imported: the two cannot drift apart) and at
    16 384 --, sampled vectors of every chunk boundary against the oracle (1e-9), every vector finite and positive."""
    import torch
    import cosmoprimo_amd as cp
    import bench
    from oracle import checks
    warnings.simplefilter('ignore')
    dev = torch.device('cuda', 0)
    n = 125000
    par = bench.eh_parameters(n, 2, torch, dev)
    host = {name: v.cpu().numpy() for name, v in par.items()}
    fid = cp.Cosmology(engine='eisenstein_hu')
    rows, rsig = {}, []
    for engine in ('wallish2018', 'brieden2022'):
        flt, parts = None, []
        for start in range(0, n, chunk):
            sl = slice(start, min(n, start + chunk))
            cosmo = cp.Cosmology(engine='eisenstein_hu', sigma8=0.8, **{name: v[sl] for name, v in par.items()})
            interp = cosmo.get_fourier().pk_interpolator(z=np.array([0.]))
            kw = dict(cosmo=cosmo, cosmo_fid=fid) if engine == 'brieden2022' else {}
            if flt is None:
                flt = cp.PowerSpectrumBAOFilter(interp, engine=engine, **kw)
            else:
                flt(interp, cosmo=cosmo if kw else None)
            parts.append(flt._pknow_rows.reshape(sl.stop - sl.start, -1).clone())
            if engine == 'wallish2018':
                rsig.append(cp.interpolator._host(cosmo._engine._rsigma8))
        rows[engine] = torch.cat(parts)
        assert rows[engine].shape == (n, 1024) and bool(torch.isfinite(rows[engine]).all()) and bool((rows[engine] > 0.).all())
    # sampled vectors: first / last of the batch and both sides of chunk boundaries.  The oracle's own find_peaks differs from the reference's run
    # by one knot that rounding decides (tests/test_oracle_bao.py); checks.brieden2022_prepared() takes the knot lists the reference itself
    # produced (golden vectors) -- nothing of the package does -- and the package's own search must have found the same lists.
    prep, _ = checks.brieden2022_prepared()
    assert all(np.array_equal(np.asarray(mine), ref) for mine, ref in zip(flt.ik_fid_peaks, prep['peaks']))
    rsig = np.concatenate(rsig)
    for i in (0, chunk - 1, chunk, (n // chunk - 1) * chunk + 17, n - 1):
        p = {name: float(v[i]) for name, v in host.items()}
        for engine in ('wallish2018', 'brieden2022'):
            ref = checks.config4_pknow(p, rsig[i], engine)
            np.testing.assert_allclose(rows[engine][i].cpu().numpy(), ref, rtol=checks.TOLERANCES['config4'], err_msg='%s %d' % (engine, i))


def test_config5_distances_full_size():
    import torch
    from cosmoprimo_amd import background
    from oracle import background as ob
    nb = 1250000
    dev = torch.device('cuda', 0)
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    t = list(bench.config5_samples(nb, 3, torch, dev))      # the bench's own draw: SURVEY.md 8(d) 5 with its w0 + wa < 1/3 rejection rule
    om, w0, wa, zz = (v.cpu().numpy() for v in t)
    assert (w0 + wa < 1. / 3.).all()
    rng = np.random.default_rng(33)

    def run():
        return background.distance('comoving_radial_distance', t[3][:, None], dict(w0_fld=t[1], wa_fld=t[2]), Omega_m=t[0], per_cosmology_z=True)[:, 0]

    out = run()
    assert out.shape == (nb,) and bool(torch.isfinite(out).all()) and torch.equal(out, run())
    host = out.cpu().numpy()
    idx = rng.integers(0, nb, 300)
    ref = np.array([ob.comoving_radial_distance(np.array([zz[i]]), ob.derived(Omega_m=om[i], w0_fld=w0[i], wa_fld=wa[i]))[0] for i in idx])
    np.testing.assert_allclose(host[idx], ref, rtol=1e-10)
    # same cosmology, growing z: distances grow; and a sample does not depend on its neighbours
    grid = torch.linspace(0., 3., 64, device=dev, dtype=torch.float64)
    first = background.distance('comoving_radial_distance', grid[None, :].expand(1000, 64).contiguous(), dict(w0_fld=t[1][:1000], wa_fld=t[2][:1000]),
                                Omega_m=t[0][:1000], per_cosmology_z=True)
    assert bool((first[:, 1:] > first[:, :-1]).all())
    alone = background.distance('comoving_radial_distance', t[3][77:78, None], dict(w0_fld=t[1][77:78], wa_fld=t[2][77:78]), Omega_m=t[0][77:78], per_cosmology_z=True)
    assert float(alone[0, 0]) == float(out[77])
