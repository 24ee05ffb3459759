"""Oracle: numpy restatement of cosmoprimo's background densities / E(z) / distance path (TEST INFRASTRUCTURE ONLY).

Follows /root/reference/cosmoprimo/cosmology.py (BaseBackground :1627-1759, distances :1855-1912,
DefaultBackground.comoving_radial_distance :2027-2042, get_default_z_interp :1940-1951, derived parameters
:355-397), cosmoprimo/jax.py (odeint rk4 :672-716, Interpolator1D natural cubic :135-196) and
cosmoprimo/constants.py.  Massless neutrinos only (m_ncdm empty, the reference default).
Vectorised over a leading batch of cosmologies.

Parity status: PINNED by tests/golden/background.npz (G8: 119-knot table, E(z), D_C/D_M/D_A/D_L from the reference).
"""
import numpy as np
from scipy import constants as sc
from scipy.interpolate import CubicSpline

# cosmoprimo/constants.py:9-21
megaparsec_over_m = 1e6 * sc.parsec
msun_over_kg = 1.98847 * 1e30
rho_crit_over_kgph_per_mph3 = 3.0 * (100. * 1e3 / megaparsec_over_m)**2 / (8 * sc.pi * sc.gravitational_constant)
rho_crit_over_Msunph_per_Mpcph3 = rho_crit_over_kgph_per_mph3 / (1e10 * msun_over_kg) * megaparsec_over_m**3
TCMB, NEFF = 2.7255, 3.044
C_KMS = sc.c / 1e3


def derived(h=0.7, Omega_cdm=0.25, Omega_b=0.05, Omega_k=0., T_cmb=TCMB, N_ur=NEFF, w0_fld=-1., wa_fld=0., Omega_m=None):
    """Density parameters the background needs (cosmology.py:355-397, 1163-1165); arrays broadcast."""
    h, Omega_b, Omega_k, T_cmb, N_ur, w0_fld, wa_fld = (np.asarray(v, dtype='f8') for v in (h, Omega_b, Omega_k, T_cmb, N_ur, w0_fld, wa_fld))
    if Omega_m is not None:   # Omega_cdm = Omega_m - Omega_b (no massive neutrinos)  cosmology.py:1163-1165
        Omega_cdm = np.asarray(Omega_m, dtype='f8') - Omega_b
    Omega_cdm = np.asarray(Omega_cdm, dtype='f8')
    rho_g = T_cmb**4 * 4. / sc.c**3 * sc.Stefan_Boltzmann                         # :356
    Omega_g = rho_g / (h**2 * rho_crit_over_kgph_per_mph3)                        # :357
    T_ur = T_cmb * (4. / 11.)**(1. / 3.)                                           # :359
    rho_ur = N_ur * 7. / 8. * T_ur**4 * 4. / sc.c**3 * sc.Stefan_Boltzmann        # :363
    Omega_ur = rho_ur / (h**2 * rho_crit_over_kgph_per_mph3)                      # :364
    Omega_de = 1. - (Omega_cdm + Omega_b + Omega_g + Omega_ur + 0. + Omega_k)     # :383 (sum in that order)
    K = -100.**2 / C_KMS**2 * Omega_k                                             # :397
    names = ['h', 'Omega_cdm', 'Omega_b', 'Omega_k', 'Omega_g', 'Omega_ur', 'Omega_de', 'w0_fld', 'wa_fld', 'K']
    vals = np.broadcast_arrays(h, Omega_cdm, Omega_b, Omega_k, Omega_g, Omega_ur, Omega_de, w0_fld, wa_fld, K)
    return dict(zip(names, vals))


def efunc(z, p):
    """E(z) = H(z)/H0 (cosmology.py:1680-1754); z broadcast against the parameter arrays."""
    rc = rho_crit_over_Msunph_per_Mpcph3
    z = np.asarray(z, dtype='f8')
    m = p['Omega_cdm'] * np.ones_like(z) * rc + p['Omega_b'] * np.ones_like(z) * rc + 0.          # :1733
    r = p['Omega_g'] * (1 + z) * rc + p['Omega_ur'] * (1 + z) * rc                                 # :1734
    de = p['Omega_de'] * (1 + z) ** (3. * (p['w0_fld'] + p['wa_fld'])) * np.exp(3. * p['wa_fld'] * (1. / (1 + z) - 1)) * rc   # :1728
    rho_crit = (m + r + de) + p['Omega_k'] / (1 + z) * rc                                          # :1736, 1705, 1749
    return np.sqrt(rho_crit * (1 + z)**3 / rc)                                                     # :1754


def densities(z, p, T_cmb=TCMB, has_fld=None):
    """BaseBackground.rho_x(z) / Omega_x(z) / T_cmb(z) for N_ncdm = 0 (cosmology.py:1680-1736, 1762-1853): dict name -> array.
    ``has_fld``: dark energy is a fluid (Cosmology._has_fld, :418-420), default (w0, wa) != (-1, 0)."""
    rc = rho_crit_over_Msunph_per_Mpcph3
    z = np.asarray(z, dtype='f8')
    if has_fld is None:
        has_fld = bool(np.any(p['w0_fld'] != -1.) or np.any(p['wa_fld'] != 0.))
    d = {}
    d['rho_g'] = p['Omega_g'] * (1 + z) * rc
    d['rho_b'] = p['Omega_b'] * np.ones_like(z) * rc
    d['rho_ur'] = p['Omega_ur'] * (1 + z) * rc
    d['rho_cdm'] = p['Omega_cdm'] * np.ones_like(z) * rc
    d['rho_k'] = p['Omega_k'] / (1 + z) * rc
    d['rho_de'] = p['Omega_de'] * (1 + z) ** (3. * (p['w0_fld'] + p['wa_fld'])) * np.exp(3. * p['wa_fld'] * (1. / (1 + z) - 1)) * rc
    d['rho_Lambda'] = (0. if has_fld else p['Omega_de']) / (1 + z)**3 * rc + 0. * z
    d['rho_fld'] = (p['Omega_de'] if has_fld else 0.) * (1 + z) ** (3. * (1 + p['w0_fld'] + p['wa_fld'])) * np.exp(3. * p['wa_fld'] * (1. / (1 + z) - 1)) * rc / (1 + z)**3
    d['rho_r'] = d['rho_g'] + d['rho_ur'] + 3. * 0.
    d['rho_m'] = d['rho_cdm'] + d['rho_b'] + 0. - 3. * 0.
    d['rho_tot'] = (d['rho_cdm'] + d['rho_b'] + 0.) + (d['rho_g'] + d['rho_ur']) + d['rho_de']
    d['rho_crit'] = d['rho_tot'] + d['rho_k']
    for name in ['g', 'b', 'ur', 'cdm', 'k', 'de', 'Lambda', 'fld', 'r', 'm']:
        d['Omega_' + name] = d['rho_' + name] / d['rho_crit']
    d['T_cmb'] = T_cmb * (1 + z)
    return d


def z_knots():
    """119 interpolation knots of comoving_radial_distance (cosmology.py:1947-1949)."""
    zm = 0.3
    return np.concatenate([np.linspace(0., zm, 20)[:-1], 1. / np.geomspace(1e-4, 1. / (1 + zm), 100)[::-1] - 1.])


def distance_table(p):
    """T_i = D_C(zc_i): the reference's RK4 with a y-independent integrand (jax.py:700-710, cosmology.py:2036-2040)."""
    zc = z_knots()
    pb = {k: np.asarray(v)[..., None] for k, v in p.items()}

    def f(z):
        return C_KMS / (100. * efunc(z, pb))

    t_last, t = zc[:-1], zc[1:]
    h = t - t_last
    k1, k2, k4 = f(t_last), f(t_last + h / 2), f(t)
    inc = h / 6. * (k1 + 2 * k2 + 2 * k2 + k4)
    # sequential accumulation y = y + inc (same rounding as the scan); first knot: h = 0 -> 0
    tab = np.concatenate([np.zeros(inc.shape[:-1] + (1,)), np.cumsum(inc, axis=-1)], axis=-1)
    return zc, tab


def comoving_radial_distance(z, p):
    """Natural cubic spline through (zc, T) (jax.py:172), NaN outside [0, zc[-1]]; z has the batch shape + trailing axes."""
    zc, tab = distance_table(p)
    z = np.asarray(z, dtype='f8')
    tab2 = tab.reshape(-1, zc.size)
    zz = np.broadcast_to(z, np.broadcast_shapes(z.shape, tab.shape[:-1] + (1,) * (z.ndim - (tab.ndim - 1)))) if tab.ndim > 1 else z
    if tab.ndim == 1:
        out = CubicSpline(zc, tab, bc_type='natural', extrapolate=False)(z)
        return np.where((z >= zc[0]) & (z <= zc[-1]), out, np.nan)
    zf = zz.reshape(tab2.shape[0], -1)
    out = np.empty_like(zf)
    for i in range(tab2.shape[0]):
        o = CubicSpline(zc, tab2[i], bc_type='natural', extrapolate=False)(zf[i])
        out[i] = np.where((zf[i] >= zc[0]) & (zf[i] <= zc[-1]), o, np.nan)
    return out.reshape(zz.shape)


GIGAYEAR_OVER_MEGAPARSEC = 3.06601394e2   # cosmoprimo/constants.py:21


def time_knots():
    """400 interpolation knots of time / age (cosmology.py:1945-1946)."""
    return 1. / np.logspace(-8, 0., 400)[::-1] - 1.


def time_table(p):
    """Cumulative integral of c / (1 + z) / (100 E) over the time knots (cosmology.py:2005-2009; RK4 == Simpson with midpoints)."""
    zc = time_knots()

    def f(z):
        return C_KMS / (1. + z) / (100. * efunc(z, p))

    t_last, t = zc[:-1], zc[1:]
    h = t - t_last
    inc = h / 6. * (f(t_last) + 2 * f(t_last + h / 2) + 2 * f(t_last + h / 2) + f(t))
    return zc, np.concatenate([[0.], np.cumsum(inc)])


def time(z, p):
    """DefaultBackground.time (cosmology.py:2000-2012), in Gyr; one cosmology; NaN outside the knots."""
    zc, tab = time_table(p)
    z = np.asarray(z, dtype='f8')
    y = (tab[-1] - tab) / p['h'] / GIGAYEAR_OVER_MEGAPARSEC
    out = CubicSpline(zc, y, bc_type='natural', extrapolate=False)(z)
    return np.where((z >= zc[0]) & (z <= zc[-1]), out, np.nan)


def age(p):
    """DefaultBackground.age (cosmology.py:2014-2025), in Gyr."""
    zc, tab = time_table(p)
    return (tab[-1] - tab[0]) / p['h'] / GIGAYEAR_OVER_MEGAPARSEC


def _sk(chi, K):
    """S_K(chi) (cosmology.py:1862-1868)."""
    K = np.broadcast_to(K, chi.shape) if np.ndim(K) else np.full(chi.shape, K)
    out = chi.copy()
    pos, neg = K > 0, K < 0
    with np.errstate(invalid='ignore'):
        out[pos] = np.sin(np.sqrt(K[pos]) * chi[pos]) / np.sqrt(K[pos])
        out[neg] = np.sinh(np.sqrt(-K[neg]) * chi[neg]) / np.sqrt(-K[neg])
    return out


def distances(z, p):
    """dict of D_C, D_M, D_A, D_L in Mpc/h (cosmology.py:1855-1912); z shape = batch shape + trailing axes."""
    dc = comoving_radial_distance(z, p)
    K = np.asarray(p['K'])
    Kb = K.reshape(K.shape + (1,) * (dc.ndim - K.ndim)) if K.ndim else K
    z = np.asarray(z, dtype='f8')
    da = _sk(dc, np.broadcast_to(Kb, dc.shape)) / (1 + z)      # :1868
    return {'comoving_radial_distance': dc, 'angular_diameter_distance': da, 'comoving_transverse_distance': da * (1. + z),
            'luminosity_distance': da * (1. + z)**2}


# ---- massive neutrinos ---------------------------------------------------------------------------
TNCDM_OVER_CMB = 0.71611                                  # cosmoprimo/constants.py:18
_EV_OVER_JOULE = sc.electron_volt
_MSUN = 1.98847 * 1e30                                     # constants.py: msun_over_kg
_MPC = 1e6 * sc.parsec


def ncdm_momenta(T_eff, m, z, out='rho'):
    """_compute_ncdm_momenta, method='laguerre' (cosmology.py:74-137): phase-space integral by 100-point Gauss-Laguerre,
    in 1e10 Msun / Mpc^3.  ``out``: 'rho' | 'p'."""
    z = np.asarray(z, dtype='f8')
    a = 1. / (1. + z)
    over_T = _EV_OVER_JOULE / (sc.Boltzmann * (T_eff / a))
    m2 = ((m * over_T)**2)[..., None]
    ti, wi = np.polynomial.laguerre.laggauss(100)
    if out == 'rho':
        f = ti**2 * np.sqrt(ti**2 + m2) / (1. + np.exp(-ti))           # :59-60, exp_sign = -1
    else:
        f = 1. / 3. * ti**4 / np.sqrt(ti**2 + m2) / (1. + np.exp(-ti))  # :65-66
    tot = np.sum(f * wi, axis=-1)
    return 7. / 8. * 4 / sc.c**3 * sc.Stefan_Boltzmann * (T_eff / a)**4 * tot / (7. * np.pi**4 / 120.) / (1e10 * _MSUN) * _MPC**3


def ncdm_knots():
    """Interpolation knots of rho_ncdm / p_ncdm (cosmology.py:1941-1943)."""
    zm = 1.
    return np.concatenate([np.linspace(0., zm, 20)[:-1], 1. / np.geomspace(1e-8, 1. / (1 + zm), 100)[::-1] - 1.])


def derived_ncdm(m_ncdm, h=0.7, Omega_cdm=0.25, Omega_b=0.05, Omega_k=0., T_cmb=TCMB, N_eff=NEFF, N_ur=None, T_ncdm_over_cmb=None,
                 w0_fld=-1., wa_fld=0., Omega_m=None):
    """``derived`` with massive species (cosmology.py:355-383, 1123-1165): N_ur from N_eff, Omega_ncdm, Omega_m -> Omega_cdm, Omega_de."""
    m_ncdm = np.atleast_1d(np.asarray(m_ncdm, dtype='f8'))
    T_over = np.full(m_ncdm.size, TNCDM_OVER_CMB if T_ncdm_over_cmb is None else T_ncdm_over_cmb, dtype='f8')
    if N_ur is None:
        N_ur = N_eff - sum(t**4 * (4. / 11.)**(-4. / 3.) for t in T_over)                  # :1127
    rc = rho_crit_over_Msunph_per_Mpcph3
    rho0 = np.array([ncdm_momenta(T_cmb * t, m, 0., 'rho') / 1.**3 / h**2 for t, m in zip(T_over, m_ncdm)])     # _get_ncdm, :441-442
    p0 = np.array([ncdm_momenta(T_cmb * t, m, 0., 'p') / 1.**3 / h**2 for t, m in zip(T_over, m_ncdm)])
    Omega_ncdm, Omega_pncdm = rho0 / rc, 3. * p0 / rc
    if Omega_m is not None:
        Omega_cdm = Omega_m - Omega_b - (sum(rho0) - 3 * sum(p0)) / rc                   # :1163-1165
    p = derived(h=h, Omega_cdm=Omega_cdm, Omega_b=Omega_b, Omega_k=Omega_k, T_cmb=T_cmb, N_ur=N_ur, w0_fld=w0_fld, wa_fld=wa_fld)
    p = {k: v for k, v in p.items()}
    p['Omega_ncdm'], p['Omega_pncdm'] = Omega_ncdm, Omega_pncdm
    p['Omega_de'] = 1. - (p['Omega_cdm'] + p['Omega_b'] + p['Omega_g'] + p['Omega_ur'] + sum(Omega_ncdm) + p['Omega_k'])      # :383
    p['N_ur'], p['m_ncdm'], p['T_ncdm'] = N_ur, m_ncdm, T_over * T_cmb
    p['T_cmb'] = T_cmb
    return p


def ncdm_tables(p):
    """DefaultBackground caches (cosmology.py:1961-1998): rho and p of every species on the knots, (nspecies, 119) each."""
    zc = ncdm_knots()
    rho = np.array([ncdm_momenta(T, m, zc, 'rho') / (1 + zc)**3 / p['h']**2 for T, m in zip(p['T_ncdm'], p['m_ncdm'])])
    pr = np.array([ncdm_momenta(T, m, zc, 'p') / (1 + zc)**3 / p['h']**2 for T, m in zip(p['T_ncdm'], p['m_ncdm'])])
    return zc, rho, pr


def ncdm_interp(p, z, out='rho'):
    """DefaultBackground.rho_ncdm / p_ncdm (interpolated, natural cubic spline, NaN outside the knots): (nspecies,) + z.shape."""
    z = np.asarray(z, dtype='f8')
    if not len(p['m_ncdm']):      # no massive species
        return np.zeros((0,) + z.shape)
    zc, rho, pr = ncdm_tables(p)
    tab = rho if out == 'rho' else pr
    val = CubicSpline(zc, tab.T, axis=0, bc_type='natural', extrapolate=False)(z.ravel()).T
    val = np.where((z.ravel() >= zc[0]) & (z.ravel() <= zc[-1]), val, np.nan)
    return val.reshape((tab.shape[0],) + z.shape)


def efunc_ncdm(z, p):
    """E(z) with massive neutrinos (cosmology.py:1731-1754 with DefaultBackground.rho_ncdm): one cosmology."""
    rc = rho_crit_over_Msunph_per_Mpcph3
    z = np.asarray(z, dtype='f8')
    ncdm = ncdm_interp(p, z, 'rho').sum(axis=0)
    m = p['Omega_cdm'] * np.ones_like(z) * rc + p['Omega_b'] * np.ones_like(z) * rc + ncdm
    r = p['Omega_g'] * (1 + z) * rc + p['Omega_ur'] * (1 + z) * rc
    de = p['Omega_de'] * (1 + z) ** (3. * (p['w0_fld'] + p['wa_fld'])) * np.exp(3. * p['wa_fld'] * (1. / (1 + z) - 1)) * rc
    rho_crit = (m + r + de) + p['Omega_k'] / (1 + z) * rc
    return np.sqrt(rho_crit * (1 + z)**3 / rc)


def comoving_radial_distance_ncdm(z, p):
    """D_C with massive neutrinos: the same scan + natural spline as comoving_radial_distance, E(z) from efunc_ncdm."""
    zc = z_knots()

    def f(zz):
        return C_KMS / (100. * efunc_ncdm(zz, p))

    t_last, t = zc[:-1], zc[1:]
    h = t - t_last
    inc = h / 6. * (f(t_last) + 2 * f(t_last + h / 2) + 2 * f(t_last + h / 2) + f(t))
    tab = np.concatenate([[0.], np.cumsum(inc)])
    z = np.asarray(z, dtype='f8')
    out = CubicSpline(zc, tab, bc_type='natural', extrapolate=False)(z)
    return np.where((z >= zc[0]) & (z <= zc[-1]), out, np.nan)


def romberg(f, a, b, divmax=15):
    """The reference's fixed-depth Romberg rule (jax.py:519-660): all ``divmax`` refinements, last entry of the last row."""
    n, intrange = 1, b - a
    ordsum = 0.5 * (f(a) + f(b))
    last_row = np.array([intrange * ordsum])
    for i in range(1, divmax + 1):
        n *= 2
        numtosum = n // 2
        h = (b - a) * 1. / numtosum
        ordsum = ordsum + np.sum(f(a + 0.5 * h + h * np.arange(numtosum)), axis=0)
        x = intrange * ordsum / n
        row = [x]
        for k, y in enumerate(last_row[:i]):
            x = (4.0**(k + 1) * x - y) / (4.0**(k + 1) - 1.0)
            row.append(x)
        last_row = np.array(row)
    return last_row[divmax]


def rs(z, p, efunc_fn=None, cosmomc=False):
    """BaseBackground.rs (cosmology.py:1914-1933), Mpc/h, or with ``cosmomc`` the sound horizon of _compute_rs_cosmomc (:202-228), proper Mpc;
    scalar z, one cosmology.  ``efunc_fn``: E(z) callable (default: no massive neutrinos)."""
    efunc_fn = efunc_fn or (lambda zz: efunc(zz, p))
    omega_b = p['Omega_b'] * p['h']**2

    def f(a):
        dtauda = 1. / (a**2 * (efunc_fn(1 / a - 1.) * (p['h'] * 100.)) / C_KMS)
        R = 3e4 * a * omega_b if cosmomc else 3 / 4. * a * p['Omega_b'] / p['Omega_g']
        return dtauda * (3 * (1 + R))**(-0.5)

    out = romberg(f, 1e-8, 1. / (1 + z))
    return out if cosmomc else out * p['h']


def zstar_cosmomc(omega_b, omega_m):
    """Hu & Sugiyama fit of the redshift of last scattering as used by CosmoMC (cosmology.py:208-210)."""
    return 1048 * (1 + 0.00124 * omega_b**(-0.738)) * (1 + (0.0783 * omega_b**(-0.238) / (1 + 39.5 * omega_b**0.763)) * omega_m**(0.560 / (1 + 21.1 * omega_b**1.81)))


def growth_ode_tables(p, mass='m', efunc_parts=None):
    """DefaultBackground.growth_factor / growth_rate caches (cosmology.py:2044-2093): RK4 (jax.py:700-710) of D'' = f2 D + f1 D' in
    eta = ln a on linspace(-6, 0, 201).  Returns (zc ascending, D, D'/D) on those knots; one cosmology without massive neutrinos."""
    rc = rho_crit_over_Msunph_per_Mpcph3

    def omegas(z):
        d = densities(z, p)
        return d['Omega_k'], d['Omega_r'], d['Omega_de'], (d['Omega_m'] if mass == 'm' else d['Omega_cdm'] + d['Omega_b'])

    def deriv(y, eta):
        z = np.exp(-eta) - 1.
        Ok, Or, Ode, Om = omegas(z)
        w_fld = p['w0_fld'] + z / (1. + z) * p['wa_fld']
        f1 = -1. - (-1. / 2. * (1. - Ok + Or + 3 * w_fld * Ode))
        f2 = 3. / 2. * Om
        return np.array([y[1], f2 * y[0] + f1 * y[1]])

    eta = np.linspace(-6., 0., 201)
    y = np.array([np.exp(eta[0]), np.exp(eta[0])])
    out, t_last = [], eta[0]
    for t in eta:
        h = t - t_last
        k1 = deriv(y, t_last)
        k2 = deriv(y + h * k1 / 2, t_last + h / 2)
        k3 = deriv(y + h * k2 / 2, t_last + h / 2)
        k4 = deriv(y + h * k3, t)
        y = y + h / 6. * (k1 + 2 * k2 + 2 * k3 + k4)
        out.append(y)
        t_last = t
    out = np.array(out)
    zc = np.exp(-eta) - 1.
    return zc[::-1], out[::-1, 0], out[::-1, 1] / out[::-1, 0]


def growth_factor_ode(z, p, mass='m', znorm=None):
    zc, D, _ = growth_ode_tables(p, mass=mass)
    spl = CubicSpline(zc, D, bc_type='natural', extrapolate=False)
    g = spl(np.asarray(z, dtype='f8'))
    return (1. + znorm) * g if znorm is not None else g / spl(0.)


def growth_rate_ode(z, p, mass='m'):
    zc, _, f = growth_ode_tables(p, mass=mass)
    return CubicSpline(zc, f, bc_type='natural', extrapolate=False)(np.asarray(z, dtype='f8'))
