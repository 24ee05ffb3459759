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
