"""Oracle: numpy restatement of cosmoprimo's analytic P(k) engines (TEST INFRASTRUCTURE ONLY).

Follows /root/reference/cosmoprimo/eisenstein_hu.py (scalars :34-92, growth :115-152, primordial :189-215,
transfer :241-283, pk_callable :321-324), eisenstein_hu_nowiggle.py (:17-51) and bbks.py (:34-64, including the
`3.89*q*(16.2*q)**2` term exactly as coded).  Vectorised over a leading batch of cosmologies:
every parameter may be an array of shape (B,), k has shape (nk,) -> outputs (B, nk) (or (nk,) for scalars).

Parity status: PINNED by tests/golden/power.npz (G7) and, with massive neutrinos, tests/golden/power_ncdm.npz.
"""
import numpy as np

from . import background as ob


def _col(v):
    v = np.asarray(v, dtype='f8')
    return v[..., None] if v.ndim else v


def eh_scalars(h, Omega_cdm, Omega_b, T_cmb=ob.TCMB):
    """EisensteinHuEngine._set_rsdrag + compute (eisenstein_hu.py:34-92), + alpha_gamma of the no-wiggle engine."""
    h, Omega_cdm, Omega_b, T_cmb = (np.asarray(v, dtype='f8') for v in (h, Omega_cdm, Omega_b, T_cmb))
    s = {}
    omega_b = s['omega_b'] = Omega_b * h**2
    omega_m = s['omega_m'] = Omega_cdm * h**2 + Omega_b * h**2
    frac_b = s['frac_b'] = omega_b / omega_m
    theta_cmb = s['theta_cmb'] = T_cmb / 2.7
    z_eq = s['z_eq'] = 2.5e4 * omega_m * theta_cmb ** (-4) - 1.
    k_eq = s['k_eq'] = 0.0746 * omega_m * theta_cmb ** (-2)
    z_drag_b1 = 0.313 * omega_m ** (-0.419) * (1 + 0.607 * omega_m ** 0.674)
    z_drag_b2 = 0.238 * omega_m ** 0.223
    z_drag = s['z_drag'] = 1345 * omega_m ** 0.251 / (1. + 0.659 * omega_m ** 0.828) * (1. + z_drag_b1 * omega_b ** z_drag_b2)
    r_drag = s['r_drag'] = 31.5 * omega_b * theta_cmb ** (-4) * (1000. / (1 + z_drag))
    r_eq = s['r_eq'] = 31.5 * omega_b * theta_cmb ** (-4) * (1000. / (1 + z_eq))
    s['rs_drag'] = 2. / (3. * k_eq) * np.sqrt(6. / r_eq) * np.log((np.sqrt(1 + r_drag) + np.sqrt(r_drag + r_eq)) / (1 + np.sqrt(r_eq)))
    s['k_silk'] = 1.6 * omega_b ** 0.52 * omega_m ** 0.73 * (1 + (10.4 * omega_m) ** (-0.95))
    a1 = (46.9 * omega_m) ** 0.670 * (1 + (32.1 * omega_m) ** (-0.532))
    a2 = (12.0 * omega_m) ** 0.424 * (1 + (45.0 * omega_m) ** (-0.582))
    s['alpha_c'] = a1 ** (-frac_b) * a2 ** (-frac_b**3)
    b1 = 0.944 / (1 + (458 * omega_m) ** (-0.708))
    b2 = 0.395 * omega_m ** (-0.0266)
    s['beta_c'] = 1. / (1 + b1 * ((1 - frac_b) ** b2) - 1)
    y_drag = (1 + z_eq) / (1 + z_drag)
    alpha_b_G = y_drag * (-6. * np.sqrt(1 + y_drag) + (2. + 3. * y_drag) * np.log((np.sqrt(1 + y_drag) + 1) / (np.sqrt(1 + y_drag) - 1)))
    s['alpha_b'] = 2.07 * k_eq * s['rs_drag'] * (1 + r_drag)**(-0.75) * alpha_b_G
    s['beta_node'] = 8.41 * omega_m ** 0.435
    s['beta_b'] = 0.5 + frac_b + (3. - 2. * frac_b) * np.sqrt((17.2 * omega_m) ** 2 + 1)
    s['alpha_gamma'] = 1. - 0.328 * np.log(431. * omega_m) * frac_b + 0.38 * np.log(22.3 * omega_m) * frac_b**2   # nowiggle :21
    return s


def transfer_eh(k, h, s):
    """Transfer.transfer_k of the full EH98 fit (eisenstein_hu.py:241-283); k in h/Mpc."""
    g = {n: _col(v) for n, v in s.items()}
    k = np.asarray(k, dtype='f8') * _col(h)
    q = k / (13.41 * g['k_eq'])
    ks = k * g['rs_drag']
    ln_beta = np.log(np.e + 1.8 * g['beta_c'] * q)
    ln_nobeta = np.log(np.e + 1.8 * q)
    C_alpha = 14.2 / g['alpha_c'] + 386. / (1 + 69.9 * q ** 1.08)
    C_noalpha = 14.2 + 386. / (1 + 69.9 * q ** 1.08)
    T_c_f = 1. / (1. + (ks / 5.4) ** 4)

    def T0(a, b):
        return a / (a + b * q**2)

    T_c = T_c_f * T0(ln_beta, C_noalpha) + (1 - T_c_f) * T0(ln_beta, C_alpha)
    s_tilde = g['rs_drag'] * (1 + (g['beta_node'] / ks)**3) ** (-1. / 3.)
    ks_tilde = k * s_tilde
    T_b_T0 = T0(ln_nobeta, C_noalpha)
    T_b_1 = T_b_T0 / (1 + (ks / 5.2)**2)
    T_b_2 = g['alpha_b'] / (1 + (g['beta_b'] / ks)**3) * np.exp(-(k / g['k_silk']) ** 1.4)
    T_b = np.sinc(ks_tilde / np.pi) * (T_b_1 + T_b_2)
    return g['frac_b'] * T_b + (1 - g['frac_b']) * T_c


def transfer_nowiggle(k, h, s):
    """eisenstein_hu_nowiggle.py:45-51."""
    g = {n: _col(v) for n, v in s.items()}
    k = np.asarray(k, dtype='f8') * _col(h)
    ks = k * g['rs_drag']
    gamma_eff = g['omega_m'] * (g['alpha_gamma'] + (1 - g['alpha_gamma']) / (1 + (0.43 * ks) ** 4))
    q = k * g['theta_cmb']**2 / gamma_eff
    L0 = np.log(2 * np.e + 1.8 * q)
    C0 = 14.2 + 731.0 / (1 + 62.5 * q)
    return L0 / (L0 + C0 * q**2)


def bbks_gamma(h, Omega_cdm, Omega_b, Omega_nu_m=0.):
    """BBKSEngine.compute (bbks.py:34-38); Omega_m = Omega_b + Omega_cdm + Omega_ncdm_tot - Omega_pncdm_tot (cosmology.py:381; ``Omega_nu_m`` = the
    last two, 0 without massive neutrinos)."""
    h, Omega_cdm, Omega_b = (np.asarray(v, dtype='f8') for v in (h, Omega_cdm, Omega_b))
    Omega_m = Omega_b + Omega_cdm + Omega_nu_m
    return Omega_m * h**2 * np.exp(-Omega_b * (1. + np.sqrt(2. * h) / Omega_m))


def transfer_bbks(k, h, gamma):
    """bbks.py:62-64, as coded."""
    q = np.asarray(k, dtype='f8') * _col(h) / _col(gamma)
    x = 2.34 * q
    return np.log(1 + x) / x * (1. + 3.89 * q * (16.2 * q)**2 + (5.47 * q)**3 + (6.71 * q)**4)**(-0.25)


def primordial_pk(k, h, A_s, n_s=0.96, alpha_s=0., beta_s=0., k_pivot=0.05):
    """Primordial.pk_k (eisenstein_hu.py:189-215); k in h/Mpc, k_pivot given in 1/Mpc."""
    h, A_s, n_s, alpha_s, beta_s = (_col(v) for v in (h, A_s, n_s, alpha_s, beta_s))
    kp = _col(k_pivot) / h
    k = np.asarray(k, dtype='f8')
    lnkkp = np.log(k / kp)
    return h**3 * A_s * (k / kp) ** (n_s - 1. + 1. / 2. * alpha_s * lnkkp + 1. / 6. * beta_s * lnkkp**2)


def A_s_fid(sigma8):
    """BaseEngine._get_A_s_fid (cosmology.py:505-510)."""
    return 2.43e-9 * (np.asarray(sigma8, dtype='f8') / 0.87659)**2


def pk_callable(k, transfer, Omega_m, h, pk_prim):
    """Fourier.pk_interpolator.pk_callable (eisenstein_hu.py:321-324): matter P(k, z=0) in (Mpc/h)^3."""
    k = np.asarray(k, dtype='f8')
    potential_to_density = (3. * _col(Omega_m) * 100**2 / (2. * ob.C_KMS**2 * k**2)) ** (-2)
    curvature_to_potential = 9. / 25. * 2. * np.pi**2 / k**3 / _col(h) ** 3
    return transfer ** 2 * potential_to_density * curvature_to_potential * pk_prim


def growth_factor(z, p, znorm=None):
    """Background.growth_factor of the analytic engines (CPT92; eisenstein_hu.py:115-140); p from oracle.background.derived."""
    rc = ob.rho_crit_over_Msunph_per_Mpcph3

    def growth(z):
        z = np.asarray(z, dtype='f8')
        pb = {n: _col(v) if np.ndim(z) else v for n, v in p.items()}
        E = ob.efunc(z, pb)
        rho_crit = E**2 * rc / (1 + z)**3   # efunc = sqrt(rho_crit (1+z)^3 / rc)
        Om = (pb['Omega_cdm'] * np.ones_like(z) * rc + pb['Omega_b'] * np.ones_like(z) * rc) / rho_crit
        Ode = pb['Omega_de'] * (1 + z) ** (3. * (pb['w0_fld'] + pb['wa_fld'])) * np.exp(3. * pb['wa_fld'] * (1. / (1 + z) - 1)) * rc / rho_crit
        return 1. / (1 + z) * 5 * Om / 2. / (Om**(4. / 7.) - Ode + (1. + Om / 2.) * (1 + Ode / 70.))

    if znorm is not None:
        return (1. + znorm) * growth(z)
    return growth(z) / growth(np.zeros_like(np.asarray(z, dtype='f8')))


def growth_rate(z, p):
    """eisenstein_hu.py:143-152."""
    rc = ob.rho_crit_over_Msunph_per_Mpcph3
    z = np.asarray(z, dtype='f8')
    pb = {n: _col(v) if np.ndim(z) else v for n, v in p.items()}
    E = ob.efunc(z, pb)
    rho_crit = E**2 * rc / (1 + z)**3
    Om = (pb['Omega_cdm'] * np.ones_like(z) * rc + pb['Omega_b'] * np.ones_like(z) * rc) / rho_crit
    wz1 = pb['w0_fld'] + (1. - 0.5) * pb['wa_fld']
    return Om**(0.55 + 0.05 * (1 + wz1))


def pk_z0(k, engine='eisenstein_hu', h=0.7, Omega_cdm=0.25, Omega_b=0.05, T_cmb=ob.TCMB, A_s=None, sigma8=0.8, n_s=0.96, alpha_s=0., beta_s=0.,
          k_pivot=0.05, rsigma8=1.):
    """P(k, z=0) of an analytic engine with A_s = A_s_fid(sigma8) * rsigma8^2 (eisenstein_hu.py:180-182) unless A_s is given."""
    A = A_s_fid(sigma8) if A_s is None else np.asarray(A_s, dtype='f8')
    A = A * np.asarray(rsigma8, dtype='f8')**2
    if engine == 'bbks':
        tr = transfer_bbks(k, h, bbks_gamma(h, Omega_cdm, Omega_b))
    else:
        s = eh_scalars(h, Omega_cdm, Omega_b, T_cmb)
        tr = transfer_eh(k, h, s) if engine == 'eisenstein_hu' else transfer_nowiggle(k, h, s)
    Omega_m = np.asarray(Omega_b, dtype='f8') + np.asarray(Omega_cdm, dtype='f8')
    return pk_callable(k, tr, Omega_m, h, primordial_pk(k, h, A, n_s, alpha_s, beta_s, k_pivot))


# ---- the analytic engines on a cosmology with massive neutrinos.  The reference computes (eisenstein_hu.py:21-33: the warnings are commented out): the
# fits take omega_cdm + omega_b (:37-38); the species enter through the background -- Omega0_m of pk_callable (:322, cosmology.py:381), Omega_m(z) and
# Omega_de(z) of the growth factor / rate (:134-135, 151-152 with cosmology.py:1704-1736), Omega_m of the BBKS shape parameter (bbks.py:38).
def _omegas_ncdm(z, p):
    """Omega_m(z), Omega_de(z) of one cosmology ``p`` = oracle.background.derived_ncdm (cosmology.py:1704-1707, 1724-1728, 1796, 1850)."""
    rc = ob.rho_crit_over_Msunph_per_Mpcph3
    z = np.asarray(z, dtype='f8')
    E = ob.efunc_ncdm(z, p)
    rho_crit = E**2 * rc / (1 + z)**3
    rho_m = p['Omega_cdm'] * np.ones_like(z) * rc + p['Omega_b'] * np.ones_like(z) * rc + ob.ncdm_interp(p, z, 'rho').sum(axis=0) \
        - 3. * ob.ncdm_interp(p, z, 'p').sum(axis=0)
    rho_de = p['Omega_de'] * (1 + z) ** (3. * (p['w0_fld'] + p['wa_fld'])) * np.exp(3. * p['wa_fld'] * (1. / (1 + z) - 1)) * rc
    return rho_m / rho_crit, rho_de / rho_crit


def growth_factor_ncdm(z, p, znorm=None):
    """Background.growth_factor (eisenstein_hu.py:115-140) of one cosmology with massive neutrinos."""
    def growth(z):
        z = np.asarray(z, dtype='f8')
        Om, Ode = _omegas_ncdm(z, p)
        return 1. / (1 + z) * 5 * Om / 2. / (Om**(4. / 7.) - Ode + (1. + Om / 2.) * (1 + Ode / 70.))

    if znorm is not None:
        return (1. + znorm) * growth(z)
    return growth(z) / growth(np.zeros(()))


def growth_rate_ncdm(z, p):
    """Background.growth_rate (eisenstein_hu.py:143-152) of one cosmology with massive neutrinos."""
    Om, _ = _omegas_ncdm(z, p)
    wz1 = p['w0_fld'] + (1. - 0.5) * p['wa_fld']
    return Om**(0.55 + 0.05 * (1 + wz1))


def pk_z0_ncdm(k, p, engine='eisenstein_hu', A_s=None, sigma8=0.8, n_s=0.96, alpha_s=0., beta_s=0., k_pivot=0.05, rsigma8=1.):
    """P(k, z=0) WITHOUT the growth factor of one cosmology with massive neutrinos: (transfer, P); ``p`` = oracle.background.derived_ncdm."""
    A = A_s_fid(sigma8) if A_s is None else np.asarray(A_s, dtype='f8')
    A = A * np.asarray(rsigma8, dtype='f8')**2
    nu_m = float(np.sum(p['Omega_ncdm']) - np.sum(p['Omega_pncdm']))
    h, Omega_cdm, Omega_b = float(p['h']), float(p['Omega_cdm']), float(p['Omega_b'])
    if engine == 'bbks':
        tr = transfer_bbks(k, h, bbks_gamma(h, Omega_cdm, Omega_b, nu_m))
    else:
        s = eh_scalars(h, Omega_cdm, Omega_b, p['T_cmb'])
        tr = transfer_eh(k, h, s) if engine == 'eisenstein_hu' else transfer_nowiggle(k, h, s)
    Omega_m = Omega_b + Omega_cdm + np.sum(p['Omega_ncdm']) - np.sum(p['Omega_pncdm'])      # cosmology.py:381
    return tr, pk_callable(k, tr, Omega_m, h, primordial_pk(k, h, A, n_s, alpha_s, beta_s, k_pivot))


# ---- eisenstein_hu_nowiggle_variants: Eisenstein & Hu 1997 (astro-ph/9710252) with massive neutrinos -------
def variants_scalars(p):
    """EisensteinHuNoWiggleVariantsEngine._set_rsdrag / compute (eisenstein_hu_nowiggle_variants.py:32-76).  ``p`` = oracle/background.py
    derived() or derived_ncdm() (+ 'T_cmb'); returns the engine's attributes as a dict."""
    h2 = p['h']**2
    s = {}
    s['omega_b'] = p['Omega_b'] * h2
    ncdm = (np.sum(p['Omega_ncdm']) - np.sum(p['Omega_pncdm'])) if 'Omega_ncdm' in p else 0.
    s['omega_m'] = p['Omega_cdm'] * h2 + p['Omega_b'] * h2 + ncdm * h2                      # :36 (omega_ncdm_tot - omega_pncdm_tot)
    s['frac_b'] = s['omega_b'] / s['omega_m']
    s['frac_cdm'] = p['Omega_cdm'] * h2 / s['omega_m']
    s['frac_cb'] = s['frac_cdm'] + s['frac_b']
    s['frac_ncdm'] = 1. - s['frac_cb']
    s['N_ncdm'] = len(p['m_ncdm']) if 'm_ncdm' in p else 0
    s['theta_cmb'] = p.get('T_cmb', ob.TCMB) / 2.7
    om, ob_ = s['omega_m'], s['omega_b']
    s['z_eq'] = 2.5e4 * om * s['theta_cmb'] ** (-4) - 1.
    s['k_eq'] = 0.0746 * om * s['theta_cmb'] ** (-2)
    b1 = 0.313 * om ** (-0.419) * (1 + 0.607 * om ** 0.674)
    b2 = 0.238 * om ** 0.223
    s['z_drag'] = 1291 * om ** 0.251 / (1. + 0.659 * om ** 0.828) * (1. + b1 * ob_ ** b2)            # 1291 here, 1345 in eisenstein_hu.py:53
    s['rs_drag'] = 44.5 * np.log(9.83 / om) / np.sqrt(1. + 10. * ob_ ** 0.75)
    fbn = s['frac_b'] + s['frac_ncdm']
    s['p_c'] = (5. - np.sqrt(1 + 24 * s['frac_cdm'])) / 4.
    s['p_cb'] = (5. - np.sqrt(1 + 24. * s['frac_cb'])) / 4.
    y_drag = (1 + s['z_eq']) / (1 + s['z_drag'])
    N = s['N_ncdm']
    alpha = s['frac_cdm'] / s['frac_cb'] * (5. - 2. * (s['p_c'] + s['p_cb'])) / (5. - 4. * s['p_cb']) * (1 + y_drag) ** (s['p_cb'] - s['p_c']) \
        * (1 + fbn * (-0.553 + 0.126 * fbn ** 2)) \
        / (1 - 0.193 * np.sqrt(s['frac_ncdm'] * N) + 0.169 * s['frac_ncdm'] * N ** 0.2) \
        * (1 + (s['p_c'] - s['p_cb']) / 2 * (1 + 1 / (3. - 4. * s['p_c']) / (7. - 4. * s['p_cb'])) / (1 + y_drag))
    s['gamma_ncdm'] = np.sqrt(alpha)
    s['beta_c'] = 1 / (1 - 0.949 * fbn)
    return s


def variants_transfer_kz(k, z, p, s, growth_k0, of='delta_m'):
    """Transfer.transfer_kz, grid=True (eisenstein_hu_nowiggle_variants.py:87-154): (nk, nz).  ``growth_k0`` = growth_factor(z, znorm=z_eq)."""
    k = np.asarray(k, dtype='f8')[:, None] * p['h']
    q = k / s['omega_m'] * s['theta_cmb'] ** 2
    N, f = s['N_ncdm'], s['frac_ncdm']
    if N:
        yfs = 17.2 * f * (1 + 0.488 * f ** (-7. / 6.)) * (N * q / f) ** 2
        t1 = growth_k0 ** (1. - s['p_cb'])
        t2 = (growth_k0 / (1 + yfs)) ** 0.7
        if of == 'delta_cb':
            growth = (1. + t2) ** (s['p_cb'] / 0.7) * t1
        else:
            growth = (s['frac_cb'] ** (0.7 / s['p_cb']) + t2) ** (s['p_cb'] / 0.7) * t1
    else:
        growth = growth_k0 = np.ones_like(np.asarray(z, dtype='f8'))
    gamma_eff = s['omega_m'] * (s['gamma_ncdm'] + (1 - s['gamma_ncdm']) / (1 + (k * s['rs_drag'] * 0.43) ** 4))
    q_eff = q * s['omega_m'] / gamma_eff
    TL = np.log(np.e + 1.84 * s['beta_c'] * s['gamma_ncdm'] * q_eff)
    TC = 14.4 + 325. / (1 + 60.5 * q_eff ** 1.08)
    T = TL / (TL + TC * q_eff ** 2)
    if N:
        qn = 3.92 * q * np.sqrt(N / f)
        T = T * (1 + 1.24 * f ** 0.64 * N ** (0.3 + 0.6 * f) / (qn ** (-1.6) + qn ** 0.8))
    return T * growth / growth_k0
