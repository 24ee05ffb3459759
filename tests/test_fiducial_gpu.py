"""Fiducial cosmologies (reference cosmoprimo/fiducial.py) end to end: the DESI fiducial (one 0.06 eV-like massive species) through the
massive-neutrino background kernels against the reference's own tabulated DESI cosmology (cosmoprimo/data/desi.dat, computed with a
Boltzmann code; 161 of its 40 002 rows kept in tests/golden/desi_table.npz) and against values produced by the reference in this container."""
import warnings

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_desi_fiducial(golden):
    import cosmoprimo_amd as cp
    from cosmoprimo_amd.fiducial import DESI, BOSS, Planck2018FullFlatLCDM, Uchuu, AbacusSummit
    warnings.simplefilter('ignore')
    g = golden('desi_table')
    cosmo = DESI()
    assert cosmo['N_ncdm'] == 1 and abs(cosmo['omega_ncdm_tot'] - 0.00064420) < 1e-12 and abs(cosmo['N_ur'] - 2.0328) < 1e-12
    ba = cosmo.get_background()
    z = g['z']
    # E(z): same physics as the Boltzmann code to 1e-6 (the radiation / neutrino content matters at high z)
    np.testing.assert_allclose(ba.efunc(z), g['efunc'], rtol=2e-6)
    # D_C: the reference's own statement for this quantity is 1e-6 for z > 0.1 and 4e-4 ... 2e-3 below (natural-spline end condition of its
    # 119-knot table); its interpolation range ends at z = 9999
    dc = ba.comoving_radial_distance(z)
    hi = (z > 0.1) & (z < 9000.)
    np.testing.assert_allclose(dc[hi], g['comoving_radial_distance'][hi], rtol=2e-6)
    lo = (z > 1e-4) & (z <= 0.1)
    np.testing.assert_allclose(dc[lo], g['comoving_radial_distance'][lo], rtol=3e-3)
    # sound horizon of the BBKS / analytic background at a drag redshift (reference tests/test_cosmology.py::test_rs): finite, ~ 147 Mpc
    rs = DESI(engine='bbks').get_background().rs(1059.94)
    assert 140. < rs / cosmo['h'] < 155.
    assert 0.0103 < cosmo['theta_cosmomc'] < 0.0105
    # P(k): sigma8 of the A_s-normalised EH97 spectrum is close to the table's 0.808
    assert abs(cosmo.get_fourier().sigma8_m / 0.807952 - 1.) < 0.05
    for factory in (BOSS, Planck2018FullFlatLCDM):
        c = factory()
        assert c['N_ncdm'] == 1 and abs(c.get_fourier().sigma8_m / c['sigma8'] - 1.) < 1e-6
    assert Uchuu('Planck2018DDE')['w0_fld'] == -0.45 and abs(Uchuu()['Omega_m'] - 0.3089) < 1e-12
    assert abs(DESI(h=0.7)['h'] - 0.7) < 1e-15
    with pytest.raises(NotImplementedError):
        AbacusSummit(name=1)
