"""GPU: the EH98 transfer function at the edges of its fast paths -- k = 0, negative, NaN and subnormal wavenumbers take the library's log (out
of line), arguments of the sine above 1e6 the library's sin -- against the oracle; everything else goes through the short log / sin / exp /
reciprocal forms, whose agreement with the operation-for-operation oracle is the subject of tests/test_cosmology_gpu.py."""
import warnings

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_transfer_edges():
    import cosmoprimo_amd as cp
    from oracle import power as opw
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        cosmo = cp.Cosmology(engine='eisenstein_hu')
        tr = cosmo.get_transfer()
        k = np.array([0., -1., np.nan, 1e-320, 1e-12, 1e-8, 1e-3, 1., 1e3, 9e3, 1.1e4, 1e5, 1e7, 1e9])
        got = np.asarray(tr.transfer_k(k))
        ba = cosmo.get_background()
        with np.errstate(all='ignore'):
            ref = opw.transfer_eh(k, cosmo['h'], opw.eh_scalars(cosmo['h'], cosmo['Omega_cdm'], cosmo['Omega_b'], cosmo['T_cmb'])).ravel()
    assert got[0] == 1. and np.isnan(got[1]) and np.isnan(got[2])
    np.testing.assert_allclose(got[3:], ref[3:], rtol=2e-11)


@pytest.mark.parametrize('engine', ['eisenstein_hu', 'eisenstein_hu_nowiggle', 'bbks'])
def test_log_k_matter(engine):
    """CP_PK_LOG_K_MATTER = log(k P(k)) formed term by term in the evaluation kernel: equal to the logarithm of the evaluated spectrum to
    rounding (the input of the sine transform of wallish2018, where both forms are used: tabulated spectra take the logarithm in the transform)."""
    from cosmoprimo_amd import power as pw
    rng = np.random.default_rng(3)
    nb = 7
    bg = dict(h=rng.uniform(0.6, 0.8, nb), Omega_cdm=rng.uniform(0.2, 0.35, nb), Omega_b=rng.uniform(0.03, 0.06, nb))
    pk = dict(A_s=rng.uniform(1e-9, 3e-9, nb), n_s=rng.uniform(0.9, 1., nb), alpha_s=rng.uniform(-0.01, 0.01, nb), beta_s=rng.uniform(-0.01, 0.01, nb))
    k = np.linspace(7e-5, 7., 4096)
    p = pw.analytic(engine, 'matter', k, bg=bg, pk=pk).cpu().numpy()
    got = pw.analytic(engine, 'log_k_matter', k, bg=bg, pk=pk).cpu().numpy()
    np.testing.assert_allclose(got, np.log(k * p), rtol=0, atol=2e-14)
    with pytest.raises(Exception):
        pw.analytic(engine, 'log_k_matter', k, z=np.array([0., 1.]), bg=bg, pk=pk)


@pytest.mark.parametrize('engine', ['eisenstein_hu', 'eisenstein_hu_nowiggle', 'bbks'])
def test_large_and_small_batches_take_the_same_values(engine):
    """cp_power_eval runs one WAVE per cosmology for batches that fill the chip and workgroups of four for small ones (the per-cosmology part of a
    thread amortised over more wavenumbers): the same arithmetic per (cosmology, k, z) -- bit for bit -- for spectra with redshifts (more of them than
    a wave has lanes), scaled wavenumbers and the term-by-term logarithm, at numbers of wavenumbers that do and do not fill the last pass of a wave."""
    import torch
    from cosmoprimo_amd import power as pw
    rng = np.random.default_rng(11)
    nb = 3200      # >= 256 x 12: the one-wave shape
    bg = dict(h=rng.uniform(0.6, 0.8, nb), Omega_cdm=rng.uniform(0.2, 0.35, nb), Omega_b=rng.uniform(0.03, 0.06, nb))
    pk = dict(A_s=rng.uniform(1e-9, 3e-9, nb), n_s=rng.uniform(0.9, 1., nb))
    few = slice(5, 40)
    sub = lambda d: {name: v[few] for name, v in d.items()}      # noqa: E731
    for nk, z, kscale in ((341, None, True), (1000, np.linspace(0., 2., 70), False), (64, None, False), (65, np.array([0.5]), False)):
        k = np.geomspace(1e-3, 5., nk)
        ks = rng.uniform(0.9, 1.1, nb) if kscale else None
        big = pw.analytic(engine, 'matter', k, z=z, bg=bg, pk=pk, kscale=ks)
        small = pw.analytic(engine, 'matter', k, z=z, bg=sub(bg), pk=sub(pk), kscale=None if ks is None else ks[few])
        assert torch.equal(big[few], small), (nk, engine)
    k = np.linspace(7e-5, 7., 777)
    assert torch.equal(pw.analytic(engine, 'log_k_matter', k, bg=bg, pk=pk)[few], pw.analytic(engine, 'log_k_matter', k, bg=sub(bg), pk=sub(pk)))
